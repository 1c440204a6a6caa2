set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_api.py -q -m gpu -x -k "far" > gpurun_out/ff_tests.log 2>&1 || { tail -40 gpurun_out/ff_tests.log; exit 1; }
tail -3 gpurun_out/ff_tests.log
for round in 1 2 3; do
for name in r03c HEAD; do
  dir=build/tree_$name; [ "$name" = HEAD ] && dir=.
  python $dir/bench.py --steps 10 --warmup 3 --no-extras --farfield 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-5s ms/step %.4f kernels %s' % ('$name', d['ms_per_step'], d['kernel_ms_per_step']))"
done
done
bash scripts/prof_farfield.sh 2>&1 | grep -v "prologue\|combine\|rocclr" | cut -c1-60,200-330
