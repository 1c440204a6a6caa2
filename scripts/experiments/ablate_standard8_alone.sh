# 8 standard-atmosphere levels per call (1013 ... 0.1 hPa), direct kernel: the accumulate launch run
# alone with parts switched off (1 general ranges, 2 fast ranges, 4 clipping lines, 16 core lines,
# 32 inner points).
# (parts of the kernel can only be switched off in the diagnostics build: python -m pylbl_amd.build ablate)
export PYLBL_AMD_LIBRARY=$(pwd)/pylbl_amd/liblbl_amd_ablate.so
for ablate in 0 1 2 3 4 16 32 48 0; do
  python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --levels-per-gpu 8 --profile standard --ablate $ablate 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ablate %-3s ms/step %.3f accumulate launch alone %.3f ms' % ('$ablate', d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
