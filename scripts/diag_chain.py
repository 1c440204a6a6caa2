import sys
import numpy as np
sys.path.insert(0, ".")
from pylbl_amd import synthetic
from pylbl_amd.engine import Engine
from oracle import oracle
e = Engine(0)
table = synthetic.line_table("CO2", 2290., 2400., num_lines=8000, seed=95, tips_range=(150, 400))
rng = np.random.default_rng(3)
near = rng.choice(table.num_lines, 1500, replace=False)
table.nu[near] = np.round(table.nu[near]) + rng.uniform(-0.004, 0.004, near.size)
table = table.subset(np.argsort(table.nu, kind="stable"))
atmos = synthetic.standard_atmosphere(6)
v0, vn, npv = 2300, 2380, 100
m = e.load(table)
res = {}
for scan in (1, 0):
    e.set_option("scan_chain", scan)
    res[scan] = e.compute(m, atmos.t, atmos.p, atmos.vmr["CO2"], v0, vn, npv, remove_pedestal=True)
plain = e.compute(m, atmos.t, atmos.p, atmos.vmr["CO2"], v0, vn, npv)
for level in range(6):
    kref, _ = oracle.absorption_port(table, atmos.t[level], atmos.p[level], atmos.vmr["CO2"][level], v0, vn, npv, remove_pedestal=True)
    d = np.abs(res[1][level] - res[0][level])/plain[level]
    d1 = np.abs(res[1][level] - kref)/plain[level]
    d0 = np.abs(res[0][level] - kref)/plain[level]
    i = int(np.argmax(d))
    print(level, "scan vs serial max", d.max(), "at", i, "cell", i//npv, "| scan vs ref", d1.max(), "serial vs ref", d0.max(), "n bad cells", len(set((np.where(d > 1e-10)[0]//npv).tolist())))
