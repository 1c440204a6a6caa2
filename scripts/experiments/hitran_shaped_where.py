"""Where the far-field path loses its gain on the HITRAN-shaped table: grids that start at 0, 1, 27 and
60 cm-1 (the lines below 1 cm-1 reach points up to 26), direct evaluations counted."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
from pylbl_amd.engine import Engine, DeviceSpectra  # noqa: E402
from tests.hitran_shapes import hitran_shaped_table  # noqa: E402

engine = Engine(0)
table = hitran_shaped_table(np.random.default_rng(8101), 1.e-4, 5026., 150_000)
handle = engine.load(table)
for v0, vn in ((0, 5000), (1, 5000), (27, 5000), (60, 5000), (0, 27), (0, 2), (2, 27)):
    out = DeviceSpectra(engine, 1, (vn - v0)*1000)
    row = []
    for farfield in (False, True):
        _, evals = engine.compute(handle, 250., 5.e4, 4.e-4, v0, vn, 1000, farfield=farfield,
                                  want_evals=True, range_policy="skip")
        for _ in range(3):
            engine.compute(handle, 250., 5.e4, 4.e-4, v0, vn, 1000, farfield=farfield, out=out,
                           asynchronous=True, range_policy="skip")
        engine.synchronize()
        start = time.perf_counter()
        for _ in range(10):
            engine.compute(handle, 250., 5.e4, 4.e-4, v0, vn, 1000, farfield=farfield, out=out,
                           asynchronous=True, range_policy="skip")
        engine.synchronize()
        row.append(((time.perf_counter() - start)/10*1e3, evals))
    print(f"grid {v0:3d}-{vn}: direct {row[0][0]:6.3f} ms ({row[0][1]:.3e} evals)   "
          f"far-field {row[1][0]:6.3f} ms ({row[1][1]:.3e} evals counted)", flush=True)
