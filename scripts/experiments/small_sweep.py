"""configs[0] and [1]: throughput of asynchronous calls into a ring of output blocks as a function of
the smallest work item (engine option item_floor) and the ring size.
Usage on the GPU box: python scripts/experiments/small_sweep.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pylbl_amd import synthetic                                     # noqa: E402
from pylbl_amd.engine import DeviceSpectra, Engine                  # noqa: E402

CASES = {"0": (("CO2",), 500., 800., 0.1), "1": (("H2O", "CO2"), 1., 5000., 0.01)}
surface = synthetic.surface_level()
for name, (mols, lo, hi, step) in CASES.items():
    v0, vn, npv = synthetic.grid_arguments(np.asarray([lo, lo + step, hi - step]))
    n = (vn - v0)*npv
    tables = [synthetic.line_table(f, lo, hi) for f in mols]
    for floor in (0, 128, 256, 512, 1024, 2048):
        for ring in (4, 8):
            engine = Engine()
            engine.set_option("item_floor", floor)
            handles = [engine.load(t) for t in tables]
            outs = [DeviceSpectra(engine, 1, n) for _ in range(ring*len(handles))]

            def step(turn):
                first = (turn % ring)*len(handles)
                for h, t, out in zip(handles, tables, outs[first:first + len(handles)]):
                    engine.compute(h, surface.t[:1], surface.p[:1], surface.vmr[t.formula][:1], v0,
                                   vn, npv, out=out, asynchronous=True)
            for turn in range(64):
                step(turn)
            engine.synchronize()
            best = []
            for _ in range(3):
                done = 0
                start = time.perf_counter()
                while time.perf_counter() - start < 0.3:      # as bench.py's lines_leg
                    for turn in range(50):
                        step(turn)
                    engine.synchronize()
                    done += 50
                best.append((time.perf_counter() - start)/done*1e3)
            print(f"config {name} item_floor {floor:5d} ring {ring}: " +
                  " ".join(f"{ms:.4f}" for ms in best) + " ms per step", flush=True)
            for out in outs:
                out.free()
            engine.close()
