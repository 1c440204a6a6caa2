"""Timing of the continuum kernels (HIP events inside the engine) on benchmark-sized grids.
Run on the GPU box:  python scripts/perf_continuum.py"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("PYLBL_MT_CKD", os.path.join(ROOT, "tests", "golden", "mt_ckd_bands.npz"))

from pylbl_amd import mt_ckd, synthetic            # noqa: E402
from pylbl_amd.engine import DeviceSpectra, default_engine   # noqa: E402

engine = default_engine(0)
cases = [("1-5000 @ 0.001", np.arange(1., 5000., 0.001)),
         ("1-5000 @ 0.0005", np.arange(1., 5000., 0.0005)),
         ("7000-60000 @ 0.01", np.arange(7000., 60000., 0.01))]
for label, grid in cases:
    for levels in (1, 16, 64):
        atmos = synthetic.standard_atmosphere(max(levels, 2))
        t, p = atmos.t[:levels], atmos.p[:levels]
        vmr = {k: v[:levels] for k, v in atmos.vmr.items()}
        out = DeviceSpectra(engine, levels, grid.size)
        for owner in ("H2OSelf", "CO2", "O2"):
            continuum = mt_ckd.CONTINUA[owner]()
            for accumulate in (False, True):
                for _ in range(2):
                    continuum.spectra_levels(t, p, vmr, grid, out=out, accumulate=accumulate)
                engine.set_option("timing", 1)
                engine.timing(reset=True)
                for _ in range(10):
                    continuum.spectra_levels(t, p, vmr, grid, out=out, accumulate=accumulate,
                                             asynchronous=True)
                engine.synchronize()
                ms, launches = engine.timing(reset=True)
                engine.set_option("timing", 0)
                per = ms[5]/launches[5]
                # Algorithmic bytes: wavenumber in, extinction out (and in again when adding),
                # per point and level.  A thread keeps its points for up to 4 levels, so the
                # kernel itself moves less: the wavenumber once per 4 levels.
                algorithmic = (24 if accumulate else 16)*grid.size*levels
                moved = ((16 if accumulate else 8)*levels + 8*((levels + 3)//4))*grid.size
                print(f"{label:18s} levels {levels:3d} {owner:8s} accumulate={int(accumulate)}: "
                      f"interp {per*1e3:8.1f} us  algorithmic {algorithmic/per/1e6:7.1f} GB/s  "
                      f"moved {moved/per/1e6:7.1f} GB/s   bands {ms[4]/launches[4]*1e3:6.1f} us",
                      flush=True)
        out.free()
