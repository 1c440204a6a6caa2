"""Spectroscopy.compute_absorption() timing with optional pre-conditions in the process (argv[1]
contains any of: torch, load, timing, pool); TREE=<path> picks the source tree.  See profiles/r03_ab_api.txt."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("TREE", "/root/repo"))
os.environ.setdefault("PYLBL_MT_CKD", "/root/repo/tests/golden/mt_ckd_bands.npz")
mode = sys.argv[1]
if "torch" in mode:
    import torch
    if "importonly" not in mode:
        torch.cuda.set_device(0)
        if "emptyonly" in mode:
            x = torch.empty(10, device="cuda")
        elif "synconly" in mode:
            torch.cuda.synchronize()
        elif "stream" in mode:
            with torch.cuda.stream(torch.cuda.Stream()):
                x = torch.zeros(10, device="cuda")
            torch.cuda.synchronize()
        elif "noalloc" not in mode:
            x = torch.zeros(10, device="cuda")
            if "freed" in mode:
                del x
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
from pylbl_amd import MemoryDatabase, Spectroscopy, synthetic
from pylbl_amd.engine import default_engine, DeviceSpectra
tables = [synthetic.line_table(f, 1., 5000.) for f in ("H2O", "CO2")]
surface = synthetic.surface_level()
level = synthetic.Atmos(p=surface.p, t=surface.t, vmr={f: surface.vmr[f] for f in ("H2O", "CO2")})
grid = np.arange(1., 5000., 0.001)
e = default_engine(0)
if "load" in mode:
    hs = [e.load(t) for t in tables]
    outs = [DeviceSpectra(e, 1, 5000000) for _ in hs]
    for _ in range(30):
        for h, t, o in zip(hs, tables, outs):
            e.compute(h, level.t, level.p, level.vmr[t.formula], 1, 5001, 1000, out=o, asynchronous=True)
    e.synchronize()
if "nullstream" in mode:
    e.order_stream_after(0)
    e.synchronize()
if "timing" in mode:
    e.set_option("timing", 2); e.timing(reset=True); e.set_option("timing", 0)
if "pool" in mode:
    block = DeviceSpectra(e, 1, grid.size); target = e.host_array((1, grid.size)); block.to_host_into(target); block.free()
spec = Spectroscopy(level, grid, MemoryDatabase(tables))
spec.delivery_pieces = int(os.environ.get("PIECES", "4"))
for fmt in ("total", "gas"):
    for _ in range(4): spec.compute_absorption(fmt)
    start = time.perf_counter()
    for _ in range(8): spec.compute_absorption(fmt)
    print(mode, fmt, (time.perf_counter()-start)/8*1e3, flush=True)
