"""Host cost of one asynchronous Engine.compute() call on a tiny grid (BASELINE configs[0]): where
the microseconds of a 30-40 us call go.  Run on the GPU box:  python scripts/perf_small_calls.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

from pylbl_amd import synthetic                                  # noqa: E402
from pylbl_amd.engine import DeviceSpectra, Engine               # noqa: E402

engine = Engine(0)
table = synthetic.line_table("CO2", 500., 800.)
handle = engine.load(table)
v0, vn, npv = 500, 801, 10
outs = [DeviceSpectra(engine, 1, (vn - v0)*npv) for _ in range(4)]
t, p, x = np.asarray([288.99]), np.asarray([98388.]), np.asarray([3.6e-4])


def burst(n):
    for i in range(n):
        engine.compute(handle, t, p, x, v0, vn, npv, out=outs[i % 4], asynchronous=True)
    engine.synchronize()


burst(200)
for n in (1000, 4000):
    start = time.perf_counter()
    burst(n)
    print(f"{n} calls: {(time.perf_counter() - start)/n*1e6:.1f} us per call", flush=True)
# Host side alone: how fast can calls be queued (the GPU drains behind)?
start = time.perf_counter()
for i in range(2000):
    engine.compute(handle, t, p, x, v0, vn, npv, out=outs[i % 4], asynchronous=True)
queued = time.perf_counter() - start
engine.synchronize()
print(f"queueing alone: {queued/2000*1e6:.1f} us per call", flush=True)
profile = cProfile.Profile()
profile.enable()
burst(2000)
profile.disable()
pstats.Stats(profile).sort_stats("tottime").print_stats(10)
