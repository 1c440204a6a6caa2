"""Does the relaxation settle every time?  The same call many times over (run under rocprofv3
--kernel-trace and look at the durations of run_chain_kernel: a few microseconds when it only reads the
flags, milliseconds when it has to do the work).  python3 scripts/experiments/chain_settles.py [calls]"""
import sys
sys.path.insert(0, ".")
from pylbl_amd import synthetic
from pylbl_amd.engine import DeviceSpectra, Engine
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 60
e = Engine(0)
tables = [synthetic.line_table(f, 1., 5000.) for f in ("H2O", "CO2")]
handles = [e.load(t) for t in tables]
outs = [DeviceSpectra(e, 1, 5_000_000) for _ in range(4)]
lev = synthetic.surface_level()
for i in range(calls):
    for j, (h, t) in enumerate(zip(handles, tables)):
        e.compute(h, lev.t, lev.p, lev.vmr[t.formula], 1, 5001, 1000, out=outs[(2*i + j) % 4],
                  remove_pedestal=True, asynchronous=True)
e.synchronize()
