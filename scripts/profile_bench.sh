#!/bin/bash
# rocprofv3 passes over the default bench command: kernel trace + stats, then the two HBM
# counters in separate passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one
# pass).  Usage (on the GPU box, from the repo root): scripts/profile_bench.sh <tag> [bench args]
# EXTRAS=continuum adds the continuum leg (bench.py --extras) to the profiled command.
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --extras ${EXTRAS:-none} "$@" > $OUT/bench_stats.json 2> $OUT/stats.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --extras ${EXTRAS:-none} "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --extras ${EXTRAS:-none} "$@" > $OUT/bench_write.json 2> $OUT/write.err || exit 1
find $OUT -name "*.csv" | head -50
