"""Wall clock of Spectroscopy.compute_absorption() over several levels of the standard atmosphere
(H2O + CO2, 5 M points, continua on, far-field series as by default) per output format, next to what
the kernels alone and the copies alone would take.  Usage on the GPU box:
python scripts/perf_api_levels.py [levels ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylbl_amd import MemoryDatabase, Spectroscopy, synthetic  # noqa: E402

os.environ.setdefault("PYLBL_MT_CKD", os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), "tests", "golden", "mt_ckd_bands.npz"))
counts = [int(x) for x in sys.argv[1:]] or [1, 4, 8, 16]
tables = [synthetic.line_table(f, 1., 5000.) for f in ("H2O", "CO2")]
grid = np.arange(1., 5000., 0.001)
for count in counts:
    full = synthetic.standard_atmosphere(count) if count > 1 else synthetic.surface_level()
    level = synthetic.Atmos(p=full.p, t=full.t, vmr={f: full.vmr[f] for f in ("H2O", "CO2")})
    spec = Spectroscopy(level, grid, MemoryDatabase(tables))
    if os.environ.get("PIECES"):
        spec.delivery_pieces = int(os.environ["PIECES"])
    row = []
    for fmt in ("total", "gas", "all"):
        for _ in range(3):
            spec.compute_absorption(fmt)
        start = time.perf_counter()
        reps = max(2, 16//count)
        for _ in range(reps):
            spec.compute_absorption(fmt)
        row.append((time.perf_counter() - start)/reps*1e3)
    mb = count*grid.size*8/1e6
    print(f"levels={count:3d}: total {row[0]:8.3f} ms ({mb:.0f} MB out), gas {row[1]:8.3f} ms "
          f"({2*mb:.0f} MB), all {row[2]:8.3f} ms ({4*mb:.0f} MB over the link)", flush=True)
    del spec
