"""Wall-clock of the user-facing call, Spectroscopy.compute_absorption(), per output format on
benchmark-sized inputs (H2O+CO2, 1-5000 cm-1 at 0.001 cm-1, synthetic line tables in a
temporary SQLite file).  Run on the GPU box:  python scripts/perf_spectroscopy.py [levels]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("PYLBL_MT_CKD", os.path.join(ROOT, "tests", "golden", "mt_ckd_bands.npz"))

from pylbl_amd import Spectroscopy, synthetic                   # noqa: E402
from pylbl_amd.database import Database, write_database         # noqa: E402

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 1
formulae = ("H2O", "CO2")
tables = [synthetic.line_table(f, 1., 5000.) for f in formulae]
with tempfile.TemporaryDirectory() as directory:
    path = os.path.join(directory, "lines.db")
    start = time.perf_counter()
    write_database(path, tables)
    print(f"database written in {time.perf_counter() - start:.1f} s", flush=True)
    full = synthetic.standard_atmosphere(max(levels, 2))
    atmos = synthetic.Atmos(p=full.p[:levels], t=full.t[:levels],
                            vmr={k: v[:levels] for k, v in full.vmr.items() if k in
                                 ("H2O", "CO2", "O2", "N2")})
    # O2 and N2 only feed the continua of H2O/CO2 here: keep them out of the gas loop.
    grid = np.arange(1., 5000., 0.001)
    spec = Spectroscopy(synthetic.Atmos(p=atmos.p, t=atmos.t,
                                        vmr={k: atmos.vmr[k] for k in formulae}),
                        grid, Database(path))
    gas = spec._molecule("CO2").gas
    for _ in range(2):
        gas.absorption_coefficient(atmos.t[0], atmos.p[0], atmos.vmr["CO2"][0], grid)
    start = time.perf_counter()
    for _ in range(5):
        k = gas.absorption_coefficient(atmos.t[0], atmos.p[0], atmos.vmr["CO2"][0], grid)
    print(f"Gas.absorption_coefficient (CO2, one level, result on the host): "
          f"{(time.perf_counter() - start)/5*1e3:.1f} ms per call", flush=True)
    del k
    for output_format in ("total", "gas", "all"):
        spec.compute_absorption(output_format=output_format)       # warm-up: uploads, plans
        start = time.perf_counter()
        repeats = 3
        for _ in range(repeats):
            out = spec.compute_absorption(output_format=output_format)
        seconds = (time.perf_counter() - start)/repeats
        print(f"{levels} level(s), format {output_format:5s}: {seconds*1e3:8.1f} ms per call "
              f"({levels/seconds:7.1f} levels/s)", flush=True)
