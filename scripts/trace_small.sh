#!/bin/bash
# Kernel timeline of the small-grid BASELINE configs (0 and 1): durations and the gaps between
# successive kernels of a step.  Usage on the GPU box: scripts/trace_small.sh <tag>
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in 0 1; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/c$c -- python3 $ROOT/bench.py --config $c --steps 20 --warmup 3 --no-extras > $OUT/bench_c$c.json 2> $OUT/c$c.err || exit 1
done
cd $ROOT
python3 - <<PY
import csv, glob
for c in (0, 1):
    f = glob.glob("$OUT/c%d/*/*_kernel_trace.csv" % c)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-80:]
    print("config", c)
    prev = None
    for r in rows[-16:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev)/1e3 if prev else 0.
        print("  %-60s dur %8.2f us  gap %8.2f us" % (r["Kernel_Name"][:60], (e - s)/1e3, gap))
        prev = e
PY
