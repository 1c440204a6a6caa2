# (parts of the kernel can only be switched off in the diagnostics build: python -m pylbl_amd.build ablate)
export PYLBL_AMD_LIBRARY=$(pwd)/pylbl_amd/liblbl_amd_ablate.so
for ablate in 0 1 2 3 4 16 32 48 0; do
  python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --farfield --ablate $ablate 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ablate %-3s ms/step %.4f accumulate launch alone %.4f ms' % ('$ablate', d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
