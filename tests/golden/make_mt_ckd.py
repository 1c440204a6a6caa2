"""Converts the reference's MT-CKD coefficient file into the fixture tests/golden/mt_ckd_bands.npz.

The coefficient file is data the reference's own test holds (tests/test_mt_ckd.py runs the 16
bands on it); /root/reference is absent on the GPU box, so the variables travel as a fixture.
Run here:  python tests/golden/make_mt_ckd.py
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from pylbl_amd import mt_ckd_data  # noqa: E402

SOURCE = "/root/reference/pyLBL/mt_ckd/mt-ckd.nc"
TARGET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mt_ckd_bands.npz")

if __name__ == "__main__":
    tables = mt_ckd_data.read_hdf5(SOURCE)
    mt_ckd_data.write_npz(TARGET, tables)
    back = mt_ckd_data.read_npz(TARGET)
    for name, table in tables.items():
        assert (back[name].data == table.data).all()
        print(f"{name:14s} {table.data.size:5d} points  {table.lower_bound:12.6f} .. "
              f"{table.upper_bound:12.6f} step {table.resolution}")
    print(TARGET, os.path.getsize(TARGET), "bytes")
