"""Where the host's time goes in Spectroscopy.compute_absorption("total") on the target workload:
when the last kernel has been queued (the call reaches engine.synchronize), when the GPU and the
copies are done, when the call returns.  GPU box: python scripts/perf_api_host.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylbl_amd import MemoryDatabase, Spectroscopy, synthetic  # noqa: E402
from pylbl_amd.engine import Engine                             # noqa: E402

os.environ.setdefault("PYLBL_MT_CKD", os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), "tests", "golden", "mt_ckd_bands.npz"))
tables = [synthetic.line_table(f, 1., 5000.) for f in ("H2O", "CO2")]
surface = synthetic.surface_level()
level = synthetic.Atmos(p=surface.p, t=surface.t, vmr={f: surface.vmr[f] for f in ("H2O", "CO2")})
grid = np.arange(1., 5000., 0.001)
spec = Spectroscopy(level, grid, MemoryDatabase(tables))
marks = {}
plain_synchronize = Engine.synchronize
plain_compute = Engine.compute
plain_many = Engine.continuum_compute_many


def synchronize(self):
    marks["queued"] = time.perf_counter()
    plain_synchronize(self)
    marks["drained"] = time.perf_counter()


def compute(self, *args, **kwargs):
    begin = time.perf_counter()
    out = plain_compute(self, *args, **kwargs)
    marks.setdefault("lines", []).append((begin, time.perf_counter()))
    return out


def many(self, *args, **kwargs):
    begin = time.perf_counter()
    out = plain_many(self, *args, **kwargs)
    marks["continua"] = (begin, time.perf_counter())
    return out


Engine.synchronize, Engine.compute, Engine.continuum_compute_many = synchronize, compute, many
for fmt in ("total", "gas"):
    for _ in range(6):
        spec.compute_absorption(fmt)
    rows = []
    for _ in range(12):
        marks.clear()
        start = time.perf_counter()
        spec.compute_absorption(fmt)
        end = time.perf_counter()
        lines = marks.get("lines", [])
        rows.append([(marks["continua"][0] - start) if "continua" in marks else 0.,
                     (marks["continua"][1] - marks["continua"][0]) if "continua" in marks else 0.]
                    + [b - a for a, b in lines]
                    + [marks["queued"] - start, marks["drained"] - marks["queued"],
                       end - marks["drained"], end - start])
    rows = np.median(np.asarray(rows), axis=0)*1e6
    names = ["before the first engine call", "continua call"] + \
        [f"lines call {i}" for i in range(len(rows) - 6)] + \
        ["everything queued at", "waiting for the GPU and the copies", "after the wait", "whole call"]
    print(fmt + ": " + "; ".join(f"{n} {v:.0f} us" for n, v in zip(names, rows)), flush=True)
