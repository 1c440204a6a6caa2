"""A table whose rows are in no order (every row its own run): the pedestal goes to the serial chain;
what the pre-pass costs on the way there.  python scripts/experiments/unsorted_table.py [lines]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pylbl_amd import synthetic
from pylbl_amd.engine import DeviceSpectra, Engine

lines = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
e = Engine(0)
table = synthetic.line_table("CO2", 1., 5000., num_lines=lines, seed=3)
order = np.random.default_rng(1).permutation(table.num_lines)
order[0] = int(np.argmin(table.nu))         # (the reference's range rule looks at the first row)
shuffled = table.subset(order)
out = DeviceSpectra(e, 1, 500_000)
for name, t in (("sorted", table), ("shuffled", shuffled)):
    h = e.load(t)
    for ped in (False, True):
        e.compute(h, 288.99, 98388., 3.6e-4, 1, 5001, 100, out=out, remove_pedestal=ped, range_policy="skip")
        t0 = time.perf_counter()
        for _ in range(3):
            e.compute(h, 288.99, 98388., 3.6e-4, 1, 5001, 100, out=out, remove_pedestal=ped, range_policy="skip")
        print(f"{name:9s} {lines} lines pedestal={ped!s:5s}: {(time.perf_counter() - t0)/3*1e3:9.3f} ms", flush=True)
    e.free(h)
