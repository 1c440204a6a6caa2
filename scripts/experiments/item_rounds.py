"""configs[1] (500 k points: 1 954 tiles for 1 792 workgroup slots): the accumulate launch of each
molecule, run alone (blocking calls, engine timing), as a function of the smallest work item
(engine option item_floor) -- how much of the launch is the last, nearly empty round of items.
Usage on the GPU box: python scripts/experiments/item_rounds.py [floors...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pylbl_amd import synthetic                                     # noqa: E402
from pylbl_amd.engine import DeviceSpectra, Engine                  # noqa: E402

floors = [int(a) for a in sys.argv[1:]] or [0, 64, 128, 256, 512, 1024, 2048, 4096]
lo, hi, step = 1., 5000., 0.01
surface = synthetic.surface_level()
v0, vn, npv = synthetic.grid_arguments(np.asarray([lo, lo + step, hi - step]))
n = (vn - v0)*npv
for formula in ("H2O", "CO2"):
    table = synthetic.line_table(formula, lo, hi)
    for floor in floors:
        engine = Engine()
        engine.set_option("item_floor", floor)
        handle = engine.load(table)
        out = DeviceSpectra(engine, 1, n)
        args = (handle, surface.t[:1], surface.p[:1], surface.vmr[formula][:1], v0, vn, npv)
        _, evals = engine.compute(*args, out=out, want_evals=True)
        for _ in range(5):
            engine.compute(*args, out=out)
        engine.set_option("timing", 1)
        engine.timing(reset=True)
        for _ in range(20):
            engine.compute(*args, out=out)
        ms, launches = engine.timing(reset=True)
        per_call = [m/20. for m in ms]
        print(f"{formula} item_floor {floor:5d}: accumulate+combine {per_call[2]*1e3:7.1f} us "
              f"({launches[2]//20} launches per call), prologue {per_call[0]*1e3:6.1f} us, "
              f"{evals/per_call[2]/1e9:.0f}e12 evals/s in the launch", flush=True)
        out.free()
        engine.close()
