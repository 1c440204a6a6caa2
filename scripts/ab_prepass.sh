#!/bin/bash
# Old pre-pass scheduling (host waits for run counts, pre-pass behind the accumulate launches)
# against the new one (bound, no wait, pre-pass first), same box.
for round in 1 2; do
for mode in "--pedestal" "--farfield --pedestal" "--banded --pedestal"; do
for opts in "--engine-option count_runs=1 --engine-option early_prepass=0" "--engine-option count_runs=0 --engine-option early_prepass=0" ""; do
  python bench.py --steps 20 --warmup 5 --no-extras $mode $opts 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('round $round %-24s %-66s ms/step %.4f' % ('$mode', '$opts', d['ms_per_step']))"
done
done
done
for opts in "--engine-option count_runs=1 --engine-option early_prepass=0" ""; do
python bench.py --steps 20 --warmup 5 --extras pedestal,farfield,api --no-cpu-baseline $opts 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
a = d['api_call']['formats']
print('%-66s pedestal_option %.3f farfield pedestal %.3f api %.3f %.3f %.3f' % ('$opts', d['pedestal_option']['ms_per_step'], d['farfield_option']['remove_pedestal']['ms_per_step'], a['total']['ms_per_call'], a['gas']['ms_per_call'], a['all']['ms_per_call']))"
done
