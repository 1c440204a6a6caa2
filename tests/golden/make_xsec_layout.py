"""Fixture: an ARTS-crossfit coefficient file in the layout the reference reads, and what the
reference's own ``CrossSection.absorption_coefficient`` makes of it.

pyLBL/arts_crossfit/cross_section.py:29-47 opens the file with ``xarray.open_dataset`` and reads
``bands`` (the band numbers), ``band<m>_fgrid`` (frequency [Hz]) and ``band<m>_coeffs``, both
transposed before use; ``calculate_xsec`` then indexes ``coeffs[i, :]`` for i < 4
(xsec_aux_functions.py:40-52), so the file stores ``band<m>_coeffs`` as [nfreq, 4].  The real
files are a download; this script writes a small synthetic one in exactly that layout
(tests/golden/xsec_layout.nc, HDF5 = netCDF-4 without the dimension scales xarray does not need
here) and runs the reference's class on it.  xarray is not installed in the image: a stand-in for
``open_dataset`` serves the arrays that were written, in the orientation they were written in --
everything after that line is the reference's own code (loaded from /root/reference by path).
Output, data only: tests/golden/xsec_layout.nc and tests/golden/xsec_layout.npz (grids, states,
expected cross sections).  Run here:  python tests/golden/make_xsec_layout.py
"""
import importlib
import os
import sys
import types

import numpy as np

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TARGET_FILE = os.path.join(HERE, "xsec_layout.nc")
TARGET = os.path.join(HERE, "xsec_layout.npz")
C0 = 299792458.0


def stored_arrays():
    """name -> array as stored.  Four bands: a plain one, one whose fit goes negative at the
    colder state (clipping + rescaling), one with exactly four frequencies (the shape that does
    not tell the orientation by itself), and one stored in descending frequency order (scipy's
    interp1d sorts its abscissa)."""
    rng = np.random.default_rng(7)
    arrays = {"bands": np.asarray([1, 2, 3, 5], dtype=np.int32)}
    for m, (lo, hi, n, offset) in zip((1, 2, 3, 5), ((700., 760., 241, 0.), (820., 835., 97, -4e-23),
                                                     (900., 903., 4, 0.), (1000., 1040., 161, 0.))):
        wavenumber = np.linspace(lo, hi, n)
        frequency = wavenumber*C0*100.
        f = np.linspace(0., 1., n)
        coeffs = np.zeros((n, 4))
        coeffs[:, 0] = 1e-22*np.exp(-((f - 0.45)/0.2)**2) + offset + 2e-24*rng.standard_normal(n)
        coeffs[:, 1] = 2e-25*rng.standard_normal(n)
        coeffs[:, 2] = 1e-28*rng.standard_normal(n)
        coeffs[:, 3] = 4e-28*rng.standard_normal(n)
        if m == 5:
            frequency, coeffs = frequency[::-1].copy(), coeffs[::-1].copy()
        arrays[f"band{m}_fgrid"] = frequency
        arrays[f"band{m}_coeffs"] = coeffs               # [nfreq, 4]
    return arrays


class _Variable(object):
    def __init__(self, data):
        self.data = data


class _Dataset(object):
    """What cross_section.py:29-40 touches of an xarray Dataset."""
    def __init__(self, arrays):
        self._arrays = arrays
        self.bands = _Variable(arrays["bands"])

    def __getitem__(self, name):
        return _Variable(self._arrays[name])

    def __enter__(self):
        return self

    def __exit__(self, *unused):
        return False


def load_reference_class(arrays):
    package = types.ModuleType("pyLBL")
    package.__path__ = [os.path.join(REFERENCE, "pyLBL")]
    sub = types.ModuleType("pyLBL.arts_crossfit")
    sub.__path__ = [os.path.join(REFERENCE, "pyLBL", "arts_crossfit")]
    stand_in = types.ModuleType("xarray")
    stand_in.open_dataset = lambda path: _Dataset(arrays)
    sys.modules.update({"pyLBL": package, "pyLBL.arts_crossfit": sub, "xarray": stand_in})
    return importlib.import_module("pyLBL.arts_crossfit.cross_section").CrossSection


def main():
    from tests import hdf5_writer
    arrays = stored_arrays()
    hdf5_writer.write(TARGET_FILE, arrays)
    cross_section = load_reference_class(arrays)("CFC11", TARGET_FILE)
    grids = {"fine": np.arange(690., 1050., 0.01), "coarse": np.arange(1., 3000., 1.),
             "knots": np.concatenate([arrays[f"band{m}_fgrid"]/(C0*100.) for m in (1, 3, 5)])}
    states = [(288.99, 98388.), (203.37, 11419.), (269.01, 117.)]
    out = {"states": np.asarray(states)}
    for name, grid in grids.items():
        out[f"grid_{name}"] = grid
        for i, (temperature, pressure) in enumerate(states):
            out[f"xsec_{name}_{i}"] = cross_section.absorption_coefficient(grid, temperature,
                                                                           pressure)
            print(name, i, "max", out[f"xsec_{name}_{i}"].max(), "nonzero",
                  int(np.count_nonzero(out[f"xsec_{name}_{i}"])))
    np.savez_compressed(TARGET, **out)
    print(TARGET_FILE, os.path.getsize(TARGET_FILE), "bytes;", TARGET, os.path.getsize(TARGET),
          "bytes")


if __name__ == "__main__":
    main()
