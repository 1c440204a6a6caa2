"""Far-field call on the grid 2-27 cm-1 of the HITRAN-shaped table with parts of the accumulate kernel
left out (engine option `ablate`: 1 general path, 2 fast ranges, 4 clipped windows, 8 left-overs of the
fast ranges, 16 core lines, 32 inner points)."""
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
v0, vn = 2, 27
out = DeviceSpectra(engine, 1, (vn - v0)*1000)
for bits in (0, 1, 2, 4, 8, 16, 32, 4 + 16, 1 + 2):
    engine.set_option("ablate", bits)
    for _ in range(3):
        engine.compute(handle, 250., 5.e4, 4.e-4, v0, vn, 1000, farfield=True, out=out,
                       asynchronous=True, range_policy="skip")
    engine.synchronize()
    start = time.perf_counter()
    for _ in range(10):
        engine.compute(handle, 250., 5.e4, 4.e-4, v0, vn, 1000, farfield=True, out=out,
                       asynchronous=True, range_policy="skip")
    engine.synchronize()
    print(f"ablate {bits:2d}: {(time.perf_counter() - start)/10*1e3:6.3f} ms", flush=True)
