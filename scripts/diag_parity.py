import sys
import numpy as np
sys.path.insert(0, ".")
from tests import golden_io
from oracle import oracle
from pylbl_amd.engine import Engine

e = Engine(0)
table, cases = golden_io.load_group("co2_band")
case = cases[3]
m = e.load(table)
_, ex = oracle.absorption_port(table, case.temperature, case.pressure, case.vmr, case.v0, case.vn, case.n_per_v, want_derived=True)
d = ex["derived"]
repwid = np.sqrt(np.log(2.))/d[:, 1]
y = repwid*d[:, 2]
for P in (1, 2, 4, 8):
    e.set_option("points_per_lane", P)
    k = e.compute(m, case.temperature, case.pressure, case.vmr, case.v0, case.vn, case.n_per_v)[0]
    rel = np.abs(k - case.k)/case.k
    worst = np.argsort(rel)[-5:][::-1]
    print("P", P, "max rel", rel.max(), "median rel", np.median(rel), "count>1e-8", (rel > 1e-8).sum())
    for i in worst:
        v = case.v0 + i/case.n_per_v
        j = np.argmin(np.abs(d[:, 0] - v))
        print(f"   i={i} v={v:.2f} rel={rel[i]:.3e} k={k[i]:.6e} ref={case.k[i]:.6e} nearest centre={d[j,0]:.5f} y={y[j]:.3f} x={(v-d[j,0])*repwid[j]:.2f}")
print("y range", y.min(), y.max(), "count y<70.55", (y < 70.55).sum())
