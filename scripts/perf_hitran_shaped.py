"""Throughput on a table of HITRAN-shaped values (tests/hitran_shapes.py) next to the uniform synthetic
one of the same size: python scripts/perf_hitran_shaped.py  (GPU box)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylbl_amd import synthetic  # noqa: E402
from pylbl_amd.engine import Engine, DeviceSpectra  # noqa: E402
from tests.hitran_shapes import hitran_shaped_table  # noqa: E402

engine = Engine(0)
tables = {
    "hitran-shaped": hitran_shaped_table(np.random.default_rng(8101), 1.e-4, 5026., 150_000),
    "uniform": synthetic.line_table("CO2", 1., 5026., num_lines=150_000, seed=5),
}
out = DeviceSpectra(engine, 1, 5_000_000)
for name, table in tables.items():
    handle = engine.load(table)
    for farfield in (False, True):
        for ped in (False, True):
            _, evals = engine.compute(handle, 250., 5.e4, 4.e-4, 0, 5000, 1000, remove_pedestal=ped,
                                      farfield=False, want_evals=True, range_policy="skip")
            for _ in range(3):
                engine.compute(handle, 250., 5.e4, 4.e-4, 0, 5000, 1000, remove_pedestal=ped,
                               farfield=farfield, out=out, asynchronous=True, range_policy="skip")
            engine.synchronize()
            start = time.perf_counter()
            for _ in range(10):
                engine.compute(handle, 250., 5.e4, 4.e-4, 0, 5000, 1000, remove_pedestal=ped,
                               farfield=farfield, out=out, asynchronous=True, range_policy="skip")
            engine.synchronize()
            ms = (time.perf_counter() - start)/10*1e3
            print(f"{name:14s} farfield={farfield!s:5s} pedestal={ped!s:5s}: {ms:7.3f} ms per spectrum, "
                  f"{float(evals)/ms/1e9:6.2f}e12 evals/s", flush=True)
    engine.free(handle)
