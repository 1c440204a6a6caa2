"""Where the time of Spectroscopy.compute_absorption() goes on the benchmark workload (1 level,
H2O + CO2, 5 M points): wall clock per output format with and without the continua, and a
cProfile of the "total" call.  Run on the GPU box:  python scripts/perf_api.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("PYLBL_MT_CKD", os.path.join(ROOT, "tests", "golden", "mt_ckd_bands.npz"))

from pylbl_amd import MemoryDatabase, Spectroscopy, synthetic       # noqa: E402

formulae = ("H2O", "CO2")
tables = [synthetic.line_table(f, 1., 5000.) for f in formulae]
surface = synthetic.surface_level()
level = synthetic.Atmos(p=surface.p, t=surface.t, vmr={f: surface.vmr[f] for f in formulae})
grid = np.arange(1., 5000., 0.001)
for continua in ("mt_ckd", None):
    for farfield in (False, True):
        spec = Spectroscopy(level, grid, MemoryDatabase(tables), continua_backend=continua,
                            farfield=farfield)
        for fmt in ("total", "gas", "all"):
            spec.compute_absorption(output_format=fmt, remove_pedestal=True)
            start = time.perf_counter()
            for _ in range(10):
                out = spec.compute_absorption(output_format=fmt, remove_pedestal=True)
            ms = (time.perf_counter() - start)/10*1e3
            print(f"continua={continua} farfield={farfield} format={fmt:5s}: {ms:7.2f} ms per call",
                  flush=True)
            del out
spec = Spectroscopy(level, grid, MemoryDatabase(tables))
spec.compute_absorption(output_format="total")
profile = cProfile.Profile()
profile.enable()
for _ in range(10):
    spec.compute_absorption(output_format="total")
profile.disable()
pstats.Stats(profile).sort_stats("cumulative").print_stats(18)
