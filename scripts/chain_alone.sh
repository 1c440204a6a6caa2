#!/bin/bash
# Durations of the pedestal pre-pass kernels when nothing runs beside them (overlap_pedestal=0:
# the pre-pass on the call's main stream, one lane) against the usual overlapped run: how much of
# the chain kernel's time under load is its own arithmetic and how much is waiting for a place.
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp; cd /tmp
for MODE in alone overlapped; do
  OUT=$ROOT/gpurun_out/chain_${TAG}_$MODE
  mkdir -p $OUT
  EXTRA=""
  if [ $MODE = alone ]; then EXTRA="--engine-option overlap_pedestal=0 --engine-option lanes=2"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --farfield --pedestal $EXTRA > $OUT/bench.json 2> $OUT/err.txt || exit 1
  python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/*/*_kernel_stats.csv")[0]
print("== $MODE: ms/step", json.load(open("$OUT/bench.json"))["ms_per_step"])
for r in csv.DictReader(open(f)):
    if any(x in r["Name"] for x in ("chain", "run_", "accumulate", "pedestal", "farfield", "fill_int")):
        print("%-46s calls %4s avg %8.1f min %8.1f max %8.1f us" % (r["Name"][:46], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
