#!/bin/bash
# Kernel timeline of configs[0] driven like bench.py's small-grid leg (asynchronous calls into a ring
# of 4 blocks): which kernels of neighbouring calls overlap.  scripts/trace_small_ring.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_ring
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/scripts/experiments/small_host.py > $OUT/run.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/t/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows)//2:len(rows)//2 + 30]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  %-28s queue %-4s start %8.2f end %8.2f dur %7.2f us" % (r["Kernel_Name"][:28], r.get("Queue_Id", "?"), (s - t0)/1e3, (e - t0)/1e3, (e - s)/1e3))
PY
tail -3 $OUT/run.log
