"""Wall clock of Spectroscopy.compute_absorption() for the target workload (1 level, H2O + CO2,
5 M points, continua on) per output format and per number of delivery pieces.
Usage on the GPU box: python scripts/perf_api.py [pieces ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylbl_amd import MemoryDatabase, Spectroscopy, synthetic  # noqa: E402

os.environ.setdefault("PYLBL_MT_CKD", os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), "tests", "golden", "mt_ckd_bands.npz"))
pieces = [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4, 6]
tables = [synthetic.line_table(f, 1., 5000.) for f in ("H2O", "CO2")]
surface = synthetic.surface_level()
level = synthetic.Atmos(p=surface.p, t=surface.t, vmr={f: surface.vmr[f] for f in
                                                        ("H2O", "CO2")})
grid = np.arange(1., 5000., 0.001)
if os.environ.get("LANES"):
    from pylbl_amd.engine import default_engine
    default_engine(0).set_option("lanes", int(os.environ["LANES"]))
if os.environ.get("POINTS_PER_LANE"):
    from pylbl_amd.engine import default_engine
    default_engine(0).set_option("points_per_lane", int(os.environ["POINTS_PER_LANE"]))
for farfield in (True, False):
    spec = Spectroscopy(level, grid, MemoryDatabase(tables), farfield=farfield)
    for count in pieces:
        spec.delivery_pieces = abs(count)
        row = []
        for fmt in ("total", "gas", "all"):
            for _ in range(4):
                spec.compute_absorption(fmt)
            start = time.perf_counter()
            for _ in range(8):
                spec.compute_absorption(fmt)
            row.append((time.perf_counter() - start)/8*1e3)
        print(f"farfield={farfield} pieces={count}: total {row[0]:.3f} ms, gas {row[1]:.3f} ms, "
              f"all {row[2]:.3f} ms", flush=True)
