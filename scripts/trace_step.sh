#!/bin/bash
# Timeline (kernels and copies, with the hardware queue each ran on) of the last timed steps of
# bench.py under rocprofv3 --kernel-trace.  Usage on the GPU box, from the repo root:
#   scripts/trace_step.sh <tag> [bench args]     -> gpurun_out/trace_step_<tag>/timeline.txt
TAG=${1:-r05}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_step_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/run.err || exit 1
cd $ROOT
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("lbl::", "").split("(")[0][:40],
                     "q%s" % r.get("Queue_Id", ""), "wg%s" % r.get("Grid_Size_X", "")))
for f in glob.glob("$OUT/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), "", ""))
rows.sort()
# the timed steps: the last burst of accumulate launches before the tail (blocking "alone" passes
# come after a gap); take everything in the 12 ms before the last gap > 1 ms that precedes them
acc = [i for i, r in enumerate(rows) if "accumulate" in r[2]]
keep = rows[max(0, len(rows) - 400):]
t0 = keep[0][0]
with open("$OUT/timeline.txt", "w") as out:
    for s, e, name, q, wg in keep:
        out.write("%9.1f -> %9.1f us (%8.1f)  %-42s %-6s %s\n" % ((s - t0)/1e3, (e - t0)/1e3, (e - s)/1e3, name, q, wg))
print(len(rows), "events; last 400 in timeline.txt")
PY
tail -c 600 $OUT/bench.json
