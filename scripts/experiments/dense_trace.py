"""Kernel times of the pedestal pre-pass on a very dense table (run under rocprofv3 --kernel-trace --stats):
python3 scripts/experiments/dense_trace.py [lines]"""
import sys
sys.path.insert(0, ".")
from pylbl_amd import synthetic
from pylbl_amd.engine import DeviceSpectra, Engine
lines = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
e = Engine(0)
e.set_option("farfield", 1)
table = synthetic.banded_line_table("CO2", 1., 5000., num_lines=lines, bands=8, seed=6, inside=True)
h = e.load(table)
out = DeviceSpectra(e, 1, 5_000_000)
for _ in range(6):
    e.compute(h, 288.99, 98388., 3.6e-4, 1, 5001, 1000, out=out, remove_pedestal=True)
