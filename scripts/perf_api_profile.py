"""cProfile of Spectroscopy.compute_absorption("total") on the target workload: where the HOST's
Python time goes (the call's first ~0.5 ms are bound by how fast the host queues).
GPU box: python scripts/perf_api_profile.py"""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylbl_amd import MemoryDatabase, Spectroscopy, synthetic  # noqa: E402

os.environ.setdefault("PYLBL_MT_CKD", os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), "tests", "golden", "mt_ckd_bands.npz"))
tables = [synthetic.line_table(f, 1., 5000.) for f in ("H2O", "CO2")]
surface = synthetic.surface_level()
level = synthetic.Atmos(p=surface.p, t=surface.t, vmr={f: surface.vmr[f] for f in ("H2O", "CO2")})
grid = np.arange(1., 5000., 0.001)
spec = Spectroscopy(level, grid, MemoryDatabase(tables))
for _ in range(10):
    spec.compute_absorption("total")
profile = cProfile.Profile()
profile.enable()
for _ in range(200):
    spec.compute_absorption("total")
profile.disable()
stats = pstats.Stats(profile)
stats.sort_stats("tottime").print_stats(28)
