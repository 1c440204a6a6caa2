#!/bin/bash
# Mean duration of the pedestal chain kernels under rocprofv3 for the far-field + pedestal step.
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/chain_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-extras --farfield --pedestal > $OUT/bench.json 2> $OUT/err.txt || exit 1
cd $ROOT
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if any(x in r["Name"] for x in ("chain", "run_sums", "accumulate", "run_links")):
        print("%-50s calls %s avg %.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3))
PY
