import sys, time
import numpy as np
sys.path.insert(0, ".")
from pylbl_amd import synthetic
from pylbl_amd.engine import DeviceSpectra, Engine
e = Engine(0)
v0, vn, npv = 1, 5001, 1000
out = DeviceSpectra(e, 1, (vn - v0)*npv)
for name, table in (("uniform", synthetic.line_table("CO2", 1., 5000.)),
                    ("banded", synthetic.banded_line_table("CO2", 1., 5000., num_lines=400_000, bands=8, seed=4))):
    h = e.load(table)
    for ped in (False, True):
        for far in (0, 1):
            e.set_option("farfield", far)
            e.compute(h, 288.99, 98388., 3.6e-4, v0, vn, npv, out=out, remove_pedestal=ped)
            e.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                e.compute(h, 288.99, 98388., 3.6e-4, v0, vn, npv, out=out, remove_pedestal=ped, asynchronous=True)
            e.synchronize()
            print(f"{name:8s} pedestal={ped!s:5s} farfield={far}: {(time.perf_counter() - t0)/5*1e3:7.3f} ms per spectrum")
    e.free(h)
