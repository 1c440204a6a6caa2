"""Which property of the HITRAN-shaped table (tests/hitran_shapes.py) costs the far-field path its
gain: the clump of lines below 1 cm-1, the zero half-widths, or the duplicated positions."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
from pylbl_amd.engine import Engine, DeviceSpectra  # noqa: E402
from tests.hitran_shapes import hitran_shaped_table  # noqa: E402

engine = Engine(0)
out = DeviceSpectra(engine, 1, 5_000_000)


def timed(table, label):
    handle = engine.load(table)
    row = []
    for farfield, ped in ((False, False), (True, False), (True, True)):
        for _ in range(3):
            engine.compute(handle, 250., 5.e4, 4.e-4, 0, 5000, 1000, remove_pedestal=ped,
                           farfield=farfield, out=out, asynchronous=True, range_policy="skip")
        engine.synchronize()
        start = time.perf_counter()
        for _ in range(10):
            engine.compute(handle, 250., 5.e4, 4.e-4, 0, 5000, 1000, remove_pedestal=ped,
                           farfield=farfield, out=out, asynchronous=True, range_policy="skip")
        engine.synchronize()
        row.append((time.perf_counter() - start)/10*1e3)
    print(f"{label:44s} direct {row[0]:6.3f}  far-field {row[1]:6.3f}  far-field+pedestal {row[2]:6.3f} ms",
          flush=True)
    engine.free(handle)


full = hitran_shaped_table(np.random.default_rng(8101), 1.e-4, 5026., 150_000)
timed(full, "as generated")
timed(full.subset(full.nu >= 1.), "without the lines below 1 cm-1")
wide = full.subset(np.ones(full.num_lines, bool))
wide.gamma_air[...] = np.maximum(wide.gamma_air, 0.01)
timed(wide, "no zero half-widths (gamma_air >= 0.01)")
calm = full.subset(np.ones(full.num_lines, bool))
calm.delta_air[...] = 0.
timed(calm, "no pressure shifts")
both = full.subset(full.nu >= 1.)
both.delta_air[...] = 0.
timed(both, "no lines below 1 cm-1, no shifts")
up = full.subset(np.ones(full.num_lines, bool))
up.delta_air[up.nu < 1.] = np.abs(up.delta_air[up.nu < 1.])
timed(up, "shifts of the lines below 1 cm-1 made positive")
few = full.subset((full.nu >= 1.) | (np.arange(full.num_lines) % 100 == 0))
timed(few, "one in a hundred of the lines below 1 cm-1")
