"""Spectroscopy.compute_absorption() per output format with the lanes of its two lines calls
shifted by 0..3: which hardware queue a copy shares with which lane used to decide 1.6 against 1.9
ms (round 4; lbl_engine::copy_streams).  Usage on the GPU box: python scripts/experiments/lane_parity.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ.setdefault("PYLBL_MT_CKD", os.path.join(os.getcwd(), "tests", "golden", "mt_ckd_bands.npz"))
from pylbl_amd import MemoryDatabase, Spectroscopy, synthetic
from pylbl_amd.engine import default_engine, DeviceSpectra
tables = [synthetic.line_table(f, 1., 5000.) for f in ("H2O", "CO2")]
surface = synthetic.surface_level()
level = synthetic.Atmos(p=surface.p, t=surface.t, vmr={f: surface.vmr[f] for f in ("H2O", "CO2")})
grid = np.arange(1., 5000., 0.001)
spec = Spectroscopy(level, grid, MemoryDatabase(tables))
e = default_engine(0)
e.set_option("skip_delivery_lanes", int(os.environ.get("SKIP", "1")))
small = e.load(synthetic.line_table("O3", 1., 200., num_lines=2000, seed=3))
blk = DeviceSpectra(e, 1, 199*100)
for fmt in ("total", "gas", "all"):
    row = []
    for shift in (0, 1, 1, 1):
        for _ in range(shift):
            e.compute(small, 250., 5e4, 1e-6, 1, 200, 100, remove_pedestal=True, out=blk, asynchronous=True)
        e.synchronize()
        for _ in range(4):
            r = spec.compute_absorption(fmt)
        t0 = time.perf_counter()
        for _ in range(10):
            r = spec.compute_absorption(fmt)
        row.append((time.perf_counter()-t0)/10*1e3)
    print(f"{fmt:5s}: " + "  ".join(f"{x:.3f}" for x in row) + " ms at four lane parities", flush=True)
