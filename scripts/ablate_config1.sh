#!/bin/bash
# configs[1] (H2O+CO2, 1-5000 @ 0.01) with parts of the accumulate kernel switched off (engine
# option "ablate": 1 general ranges, 2 fast ranges, 3 both = the kernel's fixed part, 4 clipping
# lines, 8 left-overs of the fast ranges, 16 core lines, 32 inner points, 64 clipped windows line
# by line).  Results are wrong by construction; only the times matter ("launch alone": the mean
# accumulate launch run by itself after the timed region, what profiles/r05_config1_where_the_time_goes.txt quotes).
# (parts of the kernel can only be switched off in the diagnostics build: python -m pylbl_amd.build ablate)
export PYLBL_AMD_LIBRARY=$(pwd)/pylbl_amd/liblbl_amd_ablate.so
TAG=${1:-r04}
OUT=gpurun_out/ablate_config1_$TAG.txt
: > $OUT
for extra in "--config 1" "--config 1 --points-per-lane 2" "--config 1 --points-per-lane 8"; do
  for a in 0 1 2 3 4 8 16 32 64; do
    line=$(python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline --ablate $a $extra 2>/dev/null | tail -1)
    ms=$(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.4f %.4f launch alone %.4f' % (d['ms_per_step'], d['kernel_ms_per_step']['accumulate'], d['roofline']['avg_launch_ms']))" "$line")
    echo "args='$extra' ablate=$a ms_per_step,accumulate_ms_in_step= $ms" | tee -a $OUT
  done
done
