"""configs[0] the way bench.py drives it (50 asynchronous calls into a ring of 4 blocks, then a
synchronize), for a library built with host-side timing checkpoints."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pylbl_amd import synthetic                                     # noqa: E402
from pylbl_amd.engine import DeviceSpectra, Engine                  # noqa: E402

surface = synthetic.surface_level()
v0, vn, npv = synthetic.grid_arguments(np.asarray([500., 500.1, 799.9]))
n = (vn - v0)*npv
table = synthetic.line_table("CO2", 500., 800.)
engine = Engine()
if len(sys.argv) > 1:
    engine.set_option("lanes", int(sys.argv[1]))
if len(sys.argv) > 2:
    engine.set_option("item_floor", int(sys.argv[2]))
if len(sys.argv) > 3:
    engine.set_option("graphs", int(sys.argv[3]))
handle = engine.load(table)
outs = [DeviceSpectra(engine, 1, n) for _ in range(4)]
t, p, x = surface.t[:1].copy(), surface.p[:1].copy(), surface.vmr["CO2"][:1].copy()
reference = engine.compute(handle, t, p, x, v0, vn, npv)[0].copy()
for repeat in range(3):
    done = 0
    queued = 0.
    start = time.perf_counter()
    while time.perf_counter() - start < 0.3:
        q0 = time.perf_counter()
        for turn in range(50):
            engine.compute(handle, t, p, x, v0, vn, npv, out=outs[turn % 4], asynchronous=True)
        queued += time.perf_counter() - q0
        engine.synchronize()
        done += 50
    total = time.perf_counter() - start
    print(f"{total/done*1e6:.1f} us per call, of which queueing {queued/done*1e6:.1f} us; "
          f"same bits as the blocking call: {all(np.array_equal(o.to_host()[0], reference) for o in outs)}",
          flush=True)
