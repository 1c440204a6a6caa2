cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --steps 20 --warmup 5 --no-extras --farfield --pedestal 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('farfield+pedestal ms/step %.4f' % d['ms_per_step'], d['kernel_ms_per_step'])"
python3 $R/bench.py --steps 20 --warmup 5 --no-extras --farfield --pedestal --points-per-lane 4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('farfield+pedestal P=4 ms/step %.4f' % d['ms_per_step'], d['kernel_ms_per_step'])"
python3 $R/bench.py --steps 20 --warmup 5 --no-extras --farfield --points-per-lane 4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('farfield P=4 ms/step %.4f' % d['ms_per_step'], d['kernel_ms_per_step'])"
python3 $R/bench.py --steps 20 --warmup 5 --no-extras --farfield 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('farfield P=8 ms/step %.4f' % d['ms_per_step'], d['kernel_ms_per_step'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ffped -- python3 $R/bench.py --steps 10 --warmup 3 --no-extras --farfield --pedestal > /dev/null 2>&1
f=$(find $R/gpurun_out/ffped -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | cut -c1-60,100-400 | head -14
