"""One table of HITRAN-shaped values, far-field series on, pedestal off then on: run under
rocprofv3 --kernel-trace --stats to see which kernels carry the time (scripts/perf_hitran_shaped.py
gives the wall clock)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from pylbl_amd.engine import Engine, DeviceSpectra  # noqa: E402
from tests.hitran_shapes import hitran_shaped_table  # noqa: E402

engine = Engine(0)
table = hitran_shaped_table(np.random.default_rng(8101), 1.e-4, 5026., 150_000)
handle = engine.load(table)
out = DeviceSpectra(engine, 1, 5_000_000)
ped = bool(int(os.environ.get("PED", "0")))
for _ in range(6):
    engine.compute(handle, 250., 5.e4, 4.e-4, 0, 5000, 1000, remove_pedestal=ped, farfield=True,
                   out=out, asynchronous=True, range_policy="skip")
    engine.synchronize()
