#!/bin/bash
# Timeline (kernels and copies) of the last Spectroscopy.compute_absorption("total") call of
# scripts/trace_api.py.  Usage on the GPU box: scripts/trace_api.sh <tag>
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_api_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $ROOT/scripts/trace_api.py > $OUT/run.log 2> $OUT/run.err || exit 1
cd $ROOT
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], "q%s" % r.get("Queue_Id", "")))
for f in glob.glob("$OUT/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), ""))
rows.sort()
# the last call: everything after the last gap > 20 ms
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 20e6:
        cut = i
rows = rows[cut:]
t0 = rows[0][0]
with open("$OUT/timeline.txt", "w") as out:
    for s, e, name, q in rows:
        line = "%9.1f -> %9.1f us (%8.1f)  %-50s %s" % ((s - t0)/1e3, (e - t0)/1e3, (e - s)/1e3, name, q)
        print(line)
        out.write(line + "\n")
PY
tail -3 $OUT/run.log
