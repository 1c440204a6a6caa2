"""Shapes of group_interp_kernel<PT, LV, NB> (csrc/continuum.h) on the bench's continuum
workloads: H2O foreign + self + CO2 at 5 M points x 1 level, and four continua x 16 levels at
3 M points.  Engine option interp_shape = 100 + 10 PT + LV (round 5 also swept NB = 2..4 bands at a time:
profiles/r05_perf_continuum_group.txt).  GPU box:
    python scripts/perf_continuum_group.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("PYLBL_MT_CKD", os.path.join(ROOT, "tests", "golden", "mt_ckd_bands.npz"))
from pylbl_amd import mt_ckd, synthetic                 # noqa: E402
from pylbl_amd.engine import DeviceSpectra, Engine      # noqa: E402

engine = Engine(0)
cases = (("3 continua, 1 level, 5 M points", ("H2OForeign", "H2OSelf", "CO2"), 1,
          np.arange(1., 5000., 0.001), (121, 141, 181)),
         ("6 continua, 1 level, 5 M points", ("H2OForeign", "H2OSelf", "CO2", "O3", "O2", "N2"), 1,
          np.arange(1., 5000., 0.001), (121, 141, 181)),
         ("4 continua, 16 levels, 3 M points", ("H2OForeign", "H2OSelf", "CO2", "O3"), 16,
          np.arange(1., 3000., 0.001), (124, 122, 142)))
for label, owners, levels, grid, shapes in cases:
    continua = [mt_ckd.CONTINUA[o](engine=engine) for o in owners]
    atmos = synthetic.standard_atmosphere(max(levels, 2))
    t, p = atmos.t[:levels], atmos.p[:levels]
    vmr = {k: v[:levels] for k, v in atmos.vmr.items()}
    block = DeviceSpectra(engine, levels, grid.size)
    print(label)
    for shape in shapes:
        engine.set_option("interp_shape", shape)
        for _ in range(3):
            mt_ckd.spectra_levels_many(continua, t, p, vmr, grid, block, asynchronous=True)
        engine.synchronize()
        engine.set_option("timing", 1)
        engine.timing(reset=True)
        start = time.perf_counter()
        for _ in range(20):
            mt_ckd.spectra_levels_many(continua, t, p, vmr, grid, block, asynchronous=True)
        engine.synchronize()
        wall = (time.perf_counter() - start)/20
        ms, launches = engine.timing(reset=True)
        engine.set_option("timing", 0)
        kernel = ms[5]/max(launches[5], 1)
        print(f"  PT={(shape//10)%10} LV={shape%10}: interp {kernel*1e3:7.1f} us "
              f"({grid.size*levels*16/kernel/1e6:7.1f} GB/s algorithmic), band spectra "
              f"{ms[4]/max(launches[4], 1)*1e3:5.1f} us, wall {wall*1e6:7.1f} us per call")
    engine.set_option("interp_shape", 0)
    block.free()
    del continua
