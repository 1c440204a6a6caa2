"""Spectroscopy.compute_absorption("total") for the target workload under the two orders of the
gases (Spectroscopy.total_order): "heavy_last" (the heaviest gas queued last, delivering its runs
of tiles as they finish) and "deferred" (queued first, its last kernels kept back until the others
have been queued).  GPU box: python scripts/perf_api_order.py [pieces ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylbl_amd import MemoryDatabase, Spectroscopy, synthetic  # noqa: E402

os.environ.setdefault("PYLBL_MT_CKD", os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), "tests", "golden", "mt_ckd_bands.npz"))
pieces = [int(x) for x in sys.argv[1:]] or [3, 4]
tables = [synthetic.line_table(f, 1., 5000.) for f in ("H2O", "CO2")]
surface = synthetic.surface_level()
level = synthetic.Atmos(p=surface.p, t=surface.t, vmr={f: surface.vmr[f] for f in ("H2O", "CO2")})
grid = np.arange(1., 5000., 0.001)
spec = Spectroscopy(level, grid, MemoryDatabase(tables))
from pylbl_amd.engine import default_engine  # noqa: E402
options = [x for x in os.environ.get("AB_OPTIONS", "").split(";")]      # "name=value name=value;..."
for round_ in range(3):
    for order in [o for o in os.environ.get("ORDERS", "heavy_last,deferred").split(",")]:
      for option in options:
        for pair in tuple(filter(None, option.split())):
            default_engine(0).set_option(pair.split("=")[0], int(pair.split("=")[1]))
        for count in pieces:
            spec.total_order = order
            spec.delivery_pieces = count
            for _ in range(4):
                spec.compute_absorption("total")
            times = []
            for _ in range(12):
                start = time.perf_counter()
                spec.compute_absorption("total")
                times.append(time.perf_counter() - start)
            print(f"round {round_} order={order} [{option}] pieces={count}: median {np.median(times)*1e3:.3f} ms, "
                  f"min {min(times)*1e3:.3f} ms", flush=True)
