"""What the HIP events around every accumulate launch (engine option timing = 2, what bench.py's
timed region runs with) cost the timed step: the target workload's step, `steps` at a time, with
timing 0 and 2.  GPU box: python scripts/perf_timing_overhead.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylbl_amd import synthetic                          # noqa: E402
from pylbl_amd.engine import DeviceSpectra, Engine       # noqa: E402

engine = Engine(0)
tables = [synthetic.line_table(f, 1., 5000.) for f in ("H2O", "CO2")]
handles = [engine.load(t) for t in tables]
level = synthetic.surface_level()
outs = [DeviceSpectra(engine, 1, 5_000_000) for _ in range(4)]


def step(k):
    for i, (h, t) in enumerate(zip(handles, tables)):
        engine.compute(h, level.t, level.p, level.vmr[t.formula], 1, 5001, 1000,
                       out=outs[2*(k % 2) + i], asynchronous=True)


for _ in range(5):
    step(_)
engine.synchronize()
for round_ in range(3):
    for steps in (10, 20, 100):
        for timing in (0, 2):
            engine.set_option("timing", timing)
            engine.timing(reset=True)
            engine.synchronize()
            start = time.perf_counter()
            for k in range(steps):
                step(k)
            engine.synchronize()
            elapsed = time.perf_counter() - start
            ms, launches = engine.timing(reset=True)
            print(f"round {round_} steps {steps:3d} timing {timing}: {elapsed/steps*1e3:.4f} ms/step"
                  + (f"  accumulate {ms[2]/max(launches[2], 1):.4f} ms/launch" if timing else ""),
                  flush=True)
engine.set_option("timing", 0)
