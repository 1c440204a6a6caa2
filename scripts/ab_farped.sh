#!/bin/bash
# Interleaved A/B of engine options on the far-field + pedestal step (what Spectroscopy runs by
# default), device-resident outputs.  Usage on the GPU box: scripts/ab_farped.sh <out> "<opts A>" "<opts B>" ...
# where each <opts> is a space-separated list of NAME=VALUE engine options ("" = defaults).
OUT=$1; shift
: > $OUT
for round in 1 2 3; do
  for opts in "$@"; do
    args=""
    for o in $opts; do args="$args --engine-option $o"; done
    line=$(python bench.py ${BENCH_ARGS:---farfield --pedestal} --no-extras --no-cpu-baseline --steps 40 --warmup 6 $args 2>/dev/null | tail -1)
    ms=$(echo "$line" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms/step  acc_alone %.4f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))")
    echo "round $round  [${opts:-defaults}]  $ms" | tee -a $OUT
  done
done
