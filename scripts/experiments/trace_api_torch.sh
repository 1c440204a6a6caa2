#!/bin/bash
# HIP API statistics of scripts/experiments/bisect_api.py with and without a torch kernel launched first.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in plain torch; do
  rm -rf $R/gpurun_out/api_$m
  rocprofv3 --hip-trace --stats --output-format csv -d $R/gpurun_out/api_$m -- python3 $R/scripts/experiments/bisect_api.py $m > $R/gpurun_out/api_$m.log 2>&1
  echo "== $m: $(grep total $R/gpurun_out/api_$m.log | tr '\n' ' ')"
  f=$(find $R/gpurun_out/api_$m -name "*hip_api_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 24 and r["Name"] not in ("__hipPushCallConfiguration", "__hipPopCallConfiguration", "hipGetLastError"):
        print("   %-32s calls %6s total %9.2f ms avg %9.0f ns" % (r["Name"], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])))
PY
done
