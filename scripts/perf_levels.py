"""Per-level timing of the accumulate kernel at different pressures / line distributions, all in
one process on one device (A/B numbers are only comparable within a run)."""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from pylbl_amd import synthetic
from pylbl_amd.engine import DeviceSpectra, Engine

e = Engine(0)
v0, vn, npv = 1, 5001, 1000
n = (vn - v0)*npv
out = DeviceSpectra(e, 1, n)
tables = {"CO2": synthetic.line_table("CO2", 1., 5000.),
          "H2O": synthetic.line_table("H2O", 1., 5000.),
          "CO2 banded": synthetic.banded_line_table("CO2", 1., 5000., num_lines=400_000, bands=8, seed=4)}
handles = {k: e.load(t) for k, t in tables.items()}
atmos = synthetic.standard_atmosphere(64)
e.set_option("timing", 1)


def run(name, t, p, x, reps=5, **options):
    for key, value in options.items():
        e.set_option(key, value)
    h = handles[name]
    _, evals = e.compute(h, t, p, x, v0, vn, npv, out=out, want_evals=True)
    e.timing(reset=True)
    for _ in range(reps):
        e.compute(h, t, p, x, v0, vn, npv, out=out)
    ms, launches = e.timing(reset=True)
    acc = ms[2]/launches[2]
    return acc, evals/acc*1e-9


print("level sweep (standard atmosphere), P=8")
for name in ("CO2", "H2O"):
    key = name
    for level in (0, 8, 16, 24, 32, 40, 48, 56, 63):
        acc, rate = run(name, atmos.t[level], atmos.p[level], atmos.vmr[key][level])
        print(f"  {name} level {level:2d} p={atmos.p[level]:10.2f} Pa T={atmos.t[level]:6.1f}  "
              f"accumulate {acc:7.3f} ms  {rate:6.2f} Tevals/s")
print("banded vs uniform CO2, surface level")
for name in ("CO2", "CO2 banded"):
    for P in (4, 8):
        acc, rate = run(name, 288.99, 98388., 3.6e-4, points_per_lane=P)
        print(f"  {name:11s} P={P} accumulate {acc:7.3f} ms  {rate:6.2f} Tevals/s")
e.set_option("points_per_lane", 0)
