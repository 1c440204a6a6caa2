#!/bin/bash
# Far-field series: how far a line must be from a tile (ratio x the tile's half-width) and how many terms,
# libraries built with -DLBL_FAR_RATIO / -DLBL_FAR_TERMS (build/liblbl_far_r<ratio>_t<terms>.so), interleaved;
# the truncation error of each against the direct kernel on the 5 M-point workload.
cp pylbl_amd/liblbl_amd.so /tmp/orig.so
for round in 1 2; do
for name in "$@"; do
  cp build/liblbl_far_$name.so pylbl_amd/liblbl_amd.so
  for args in "--farfield" "--farfield --pedestal"; do
  python bench.py --steps 20 --warmup 5 --no-extras $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-10s round $round %-22s ms/step %.4f' % ('$name', '$args', d['ms_per_step']))"
  done
done
done
for name in "$@"; do
  cp build/liblbl_far_$name.so pylbl_amd/liblbl_amd.so
  python - <<PY
import numpy as np, sys
sys.path.insert(0, ".")
from pylbl_amd import synthetic
from pylbl_amd.engine import Engine
e = Engine(0)
worst = 0.
for f in ("H2O", "CO2"):
    t = synthetic.line_table(f, 1., 5000.)
    h = e.load(t)
    lev = synthetic.surface_level()
    a = e.compute(h, lev.t, lev.p, lev.vmr[f], 1, 5001, 1000)
    b = e.compute(h, lev.t, lev.p, lev.vmr[f], 1, 5001, 1000, farfield=True)
    nz = a != 0
    worst = max(worst, float(np.max(np.abs(b[nz] - a[nz])/a[nz])))
print("$name max relative difference to the direct kernel: %.3g" % worst)
PY
done
cp /tmp/orig.so pylbl_amd/liblbl_amd.so
