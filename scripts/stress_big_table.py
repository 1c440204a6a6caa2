"""Stress: one molecule with 4 M lines on the 5 M-point grid (2e11 evals per spectrum)."""
import sys, time
sys.path.insert(0, ".")
from pylbl_amd import synthetic
from pylbl_amd.engine import DeviceSpectra, Engine
e = Engine(0)
v0, vn, npv = 1, 5001, 1000
out = DeviceSpectra(e, 1, (vn - v0)*npv)
t0 = time.perf_counter()
table = synthetic.banded_line_table("CO2", 1., 5000., num_lines=4_000_000, bands=12, seed=9)
h = e.load(table)
print(f"table + upload {time.perf_counter() - t0:.2f} s")
for far in (0, 1):
    e.set_option("farfield", far)
    for ped in (False, True):
        _, evals = e.compute(h, 288.99, 98388., 3.6e-4, v0, vn, npv, out=out, remove_pedestal=ped, want_evals=True)
        t0 = time.perf_counter()
        for _ in range(3):
            e.compute(h, 288.99, 98388., 3.6e-4, v0, vn, npv, out=out, remove_pedestal=ped)
        dt = (time.perf_counter() - t0)/3
        k = out.to_host()
        print(f"farfield={far} pedestal={ped!s:5s}: {dt*1e3:8.2f} ms  {evals/dt:.3e} evals/s  finite={bool((k == k).all())} min={k.min():.3e}")
