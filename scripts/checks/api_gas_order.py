import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("PYLBL_MT_CKD", os.path.join(sys.path[0], "tests", "golden", "mt_ckd_bands.npz"))
from pylbl_amd import MemoryDatabase, Spectroscopy, synthetic
tables = [synthetic.line_table(f, 1., 5000.) for f in ("H2O", "CO2")]
s = synthetic.surface_level()
grid = np.arange(1., 5000., 0.001)
for order in (("H2O", "CO2"), ("CO2", "H2O")):
    level = synthetic.Atmos(p=s.p, t=s.t, vmr={f: s.vmr[f] for f in order})
    spec = Spectroscopy(level, grid, MemoryDatabase(tables))
    for fmt in ("total", "gas"):
        for _ in range(5): spec.compute_absorption(fmt)
        t = time.perf_counter()
        for _ in range(20): spec.compute_absorption(fmt)
        print(f"gases {order} {fmt}: {(time.perf_counter()-t)/20*1e3:.2f} ms")
