"""Timeline of one Spectroscopy.compute_absorption("total") call under rocprofv3 --kernel-trace
--memory-copy-trace: prints the kernels and copies of the last call with their start/end relative
to the call's first kernel.  Run on the GPU box:
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d out -- python3 scripts/trace_api.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("PYLBL_MT_CKD", os.path.join(ROOT, "tests", "golden", "mt_ckd_bands.npz"))

from pylbl_amd import MemoryDatabase, Spectroscopy, synthetic       # noqa: E402

formulae = ("H2O", "CO2")
tables = [synthetic.line_table(f, 1., 5000.) for f in formulae]
levels = int(os.environ.get("LEVELS", "1"))
surface = synthetic.standard_atmosphere(levels) if levels > 1 else synthetic.surface_level()
level = synthetic.Atmos(p=surface.p, t=surface.t, vmr={f: surface.vmr[f] for f in formulae})
fmt = os.environ.get("FORMAT", "total")
grid = np.arange(1., 5000., 0.001)
spec = Spectroscopy(level, grid, MemoryDatabase(tables))
spec.delivery_pieces = int(os.environ.get("PIECES", "4"))
for _ in range(6):
    spec.compute_absorption(output_format=fmt)
time.sleep(0.05)
start = time.perf_counter()
spec.compute_absorption(output_format=fmt)
print(f"last call: {(time.perf_counter() - start)*1e3:.2f} ms")
