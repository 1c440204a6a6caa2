"""Spectroscopy.compute_absorption("total"): the heaviest gas queued first and finished last
(LBL_DEFER_FINISH, round 3) against queued last and delivering piece by piece (round 4), in one
process, alternating.  Usage on the GPU box: python scripts/experiments/ab_total_order.py [pieces ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pylbl_amd import MemoryDatabase, Spectroscopy, synthetic  # noqa: E402

os.environ.setdefault("PYLBL_MT_CKD", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))), "tests", "golden", "mt_ckd_bands.npz"))
pieces = [int(x) for x in sys.argv[1:]] or [2, 4, 6, 8]
tables = [synthetic.line_table(f, 1., 5000.) for f in ("H2O", "CO2")]
surface = synthetic.surface_level()
level = synthetic.Atmos(p=surface.p, t=surface.t, vmr={f: surface.vmr[f] for f in
                                                        ("H2O", "CO2")})
grid = np.arange(1., 5000., 0.001)
spec = Spectroscopy(level, grid, MemoryDatabase(tables))
reference = None
for count in pieces:
    spec.delivery_pieces = count
    for round_ in range(2):
        for order in ("deferred", "heavy_last"):
            spec.total_order = order
            for _ in range(3):
                out = spec.compute_absorption("total")
            start = time.perf_counter()
            for _ in range(10):
                out = spec.compute_absorption("total")
            ms = (time.perf_counter() - start)/10*1e3
            values = np.array(out["absorption"])
            if reference is None:
                reference = values
            worst = float(np.max(np.abs(values - reference)/np.maximum(np.abs(reference), 1e-300)))
            print(f"pieces={count} order={order:10s}: {ms:.3f} ms per call "
                  f"(max relative difference to the first result {worst:.2e})", flush=True)
