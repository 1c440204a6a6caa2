import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
ROOT = sys.path[0]
os.environ.setdefault("PYLBL_MT_CKD", os.path.join(ROOT, "tests", "golden", "mt_ckd_bands.npz"))
from pylbl_amd import mt_ckd, synthetic
from pylbl_amd.engine import DeviceSpectra, Engine
engine = Engine(0)
owners = ("H2OForeign", "H2OSelf", "CO2")
continua = [mt_ckd.CONTINUA[o](engine=engine) for o in owners]
atmos = synthetic.standard_atmosphere(2)
t, p = atmos.t[:1], atmos.p[:1]
vmr = {k: v[:1] for k, v in atmos.vmr.items()}
exact = np.arange(1., 5000., 0.001)
moved = exact.copy(); moved[12345] = np.nextafter(moved[12345], np.inf)
for name, grid in (("arange", exact), ("one ulp off", moved)):
    block = DeviceSpectra(engine, 1, grid.size)
    for members, label in ((continua, "3 in one pass"), (continua[2:], "CO2 alone"), (continua[:1], "H2OForeign alone")):
        for shape in (0,):
            for _ in range(3):
                mt_ckd.spectra_levels_many(members, t, p, vmr, grid, block, asynchronous=True)
            engine.synchronize()
            engine.set_option("timing", 1); engine.timing(reset=True)
            for _ in range(20):
                mt_ckd.spectra_levels_many(members, t, p, vmr, grid, block, asynchronous=True)
            engine.synchronize()
            ms, launches = engine.timing(reset=True); engine.set_option("timing", 0)
            print(f"{name:12s} {label:18s}: interp {ms[5]/launches[5]*1e3:6.1f} us", flush=True)
    block.free()
