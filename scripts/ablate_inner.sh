#!/bin/bash
# Where the general path's time goes after the inner points moved into packed passes (engine
# option "ablate": 1 no general ranges, 16 no core lines, 32 no inner points, 4 no clipping
# lines; results are wrong by construction).  Usage on the GPU box: scripts/ablate_inner.sh <tag>
# (parts of the kernel can only be switched off in the diagnostics build: python -m pylbl_amd.build ablate)
export PYLBL_AMD_LIBRARY=$(pwd)/pylbl_amd/liblbl_amd_ablate.so
TAG=${1:-r03}
OUT=gpurun_out/ablate_inner_$TAG.txt
: > $OUT
for extra in "" "--config 1" "--levels-per-gpu 8 --profile standard"; do
  for a in 0 32 16 4 1 0; do
    line=$(python bench.py --steps 20 --warmup 3 --no-extras --ablate $a $extra 2>/dev/null | tail -1)
    ms=$(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.4f %.4f' % (d['ms_per_step'], d['kernel_ms_per_step']['accumulate']))" "$line")
    echo "args='$extra' ablate=$a ms_per_step,accumulate_ms= $ms" | tee -a $OUT
  done
done
