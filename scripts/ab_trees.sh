#!/bin/bash
# A/B of whole source trees (build/tree_<name>/, each with its own built library) against the
# working tree on several workloads, interleaved; compares the event-timed accumulate kernel time
# per step, which does not depend on how each tree's bench.py drives the timed loop.
# Usage on the GPU box: scripts/ab_trees.sh r01 [more tree names]
for round in $(seq 1 ${ROUNDS:-2}); do
for args in "--config target" "--pedestal" "--banded" "--config 1" "--config 2" "--levels-per-gpu 8 --profile standard" "--farfield"; do
for name in "$@" HEAD; do
  dir=build/tree_$name
  [ "$name" = HEAD ] && dir=.
  python $dir/bench.py --steps ${STEPS:-10} --warmup 3 --no-extras $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-5s round $round %-42s ms/step %.4f accumulate ms/step %.4f' % ('$name', '$args', d['ms_per_step'], d['kernel_ms_per_step']['accumulate']))"
done
done
done
