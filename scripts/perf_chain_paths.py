"""Pedestal pass time with the parallel chain (relaxation) on/off for uniform and banded tables (one process)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pylbl_amd import synthetic
from pylbl_amd.engine import DeviceSpectra, Engine
e = Engine(0)
e.set_option("farfield", 1)
v0, vn, npv = 1, 5001, 1000
out = DeviceSpectra(e, 1, (vn - v0)*npv)
tables = {"uniform 400k": synthetic.line_table("CO2", 1., 5000.),
          "banded 400k": synthetic.banded_line_table("CO2", 1., 5000., num_lines=400_000, bands=8, seed=4),
          "banded 1.6M": synthetic.banded_line_table("CO2", 1., 5000., num_lines=1_600_000, bands=8, seed=5)}
for name, table in tables.items():
    h = e.load(table)
    for scan in (1, 0):
        e.set_option("scan_chain", scan)
        for ped in (False, True):
            e.compute(h, 288.99, 98388., 3.6e-4, v0, vn, npv, out=out, remove_pedestal=ped)
            e.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                e.compute(h, 288.99, 98388., 3.6e-4, v0, vn, npv, out=out, remove_pedestal=ped)
            dt = (time.perf_counter() - t0)/5*1e3
            print(f"{name:13s} scan_chain={scan} pedestal={ped!s:5s}: {dt:7.3f} ms per spectrum (blocking calls)")
    e.free(h)
