#!/bin/bash
# Where the general path's time goes: the default workload (and two others) with parts of the
# accumulate kernel switched off (engine option "ablate": 1 general ranges, 2 fast ranges,
# 4 clipping lines, 8 left-overs of the fast ranges, 16 core lines).  Results are wrong by
# construction; only ms_per_step matters.  Usage on the GPU box: scripts/ablate_general.sh <tag>
# (parts of the kernel can only be switched off in the diagnostics build: python -m pylbl_amd.build ablate)
export PYLBL_AMD_LIBRARY=$(pwd)/pylbl_amd/liblbl_amd_ablate.so
TAG=${1:-r02}
OUT=gpurun_out/ablate_$TAG.txt
: > $OUT
for extra in "" "--config 1" "--levels-per-gpu 8 --profile standard" "--farfield"; do
  for a in 0 1 2 4 8 16 28; do
    line=$(python bench.py --steps 20 --warmup 3 --no-extras --ablate $a $extra 2>/dev/null | tail -1)
    ms=$(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.4f %.4f' % (d['ms_per_step'], d['kernel_ms_per_step']['accumulate']))" "$line")
    echo "args='$extra' ablate=$a ms_per_step,accumulate_ms= $ms" | tee -a $OUT
  done
done
