#!/bin/bash
# HIP API time of asynchronous calls on configs[0] (scripts/perf_small_calls.py) by API name.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --hip-trace --stats --output-format csv -d $R/gpurun_out/small_api -- python3 $R/scripts/perf_small_calls.py > $R/gpurun_out/small_api.log 2>&1
f=$(find $R/gpurun_out/small_api -name "*hip_api_stats.csv" | head -1)
head -25 $f
