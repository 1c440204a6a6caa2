#!/bin/bash
# A/B of source trees on the small-grid BASELINE configs (build/tree_<name>/ against the working
# tree), interleaved: the timed line and the 4-calls-in-flight leg of bench.py.
for round in 1 2; do
for name in "$@" HEAD; do
  dir=build/tree_$name
  [ "$name" = HEAD ] && dir=.
  for c in 0 1; do
    python $dir/bench.py --steps 200 --warmup 20 --config $c --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-5s round $round config $c timed line: ms/step %.4f  evals/s %.4g' % ('$name', d['ms_per_step'], d['value']))"
  done
  python $dir/bench.py --steps 5 --warmup 2 --extras small --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())['small_grid_options']
print('%-5s round $round in flight: config0 %.4f ms %.4g evals/s   config1 %.4f ms %.4g evals/s' % ('$name', d['config0']['ms_per_step'], d['config0']['value'], d['config1']['ms_per_step'], d['config1']['value']))"
done
done
