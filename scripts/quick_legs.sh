#!/bin/bash
# Three workloads the general path matters for, timed steps only: default (target), configs[1],
# 8 standard-atmosphere levels (plain and with the pedestal).  Usage: scripts/quick_legs.sh <tag>
TAG=${1:-x}
OUT=gpurun_out/quick_$TAG.txt
: > $OUT
for extra in "" "--config 1" "--levels-per-gpu 8 --profile standard" "--levels-per-gpu 8 --profile standard --pedestal" "--farfield"; do
  for rep in 1 2; do
    line=$(python bench.py --steps 20 --warmup 3 --no-extras $extra 2>/dev/null | tail -1)
    ms=$(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.4f %.4f %.4g' % (d['ms_per_step'], d['kernel_ms_per_step']['accumulate'], d['value']))" "$line")
    echo "args='$extra' ms_per_step,accumulate_ms,evals/s= $ms" | tee -a $OUT
  done
done
