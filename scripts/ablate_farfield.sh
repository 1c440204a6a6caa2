#!/bin/bash
# Which part of the far-field step is what: the diagnostic switches of accumulate_tile on --farfield.
# (parts of the kernel can only be switched off in the diagnostics build: python -m pylbl_amd.build ablate)
export PYLBL_AMD_LIBRARY=$(pwd)/pylbl_amd/liblbl_amd_ablate.so
for ablate in 0 1 2 3 4 8 16 32 0; do
  python bench.py --steps 10 --warmup 3 --no-extras --farfield --ablate $ablate 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ablate %-3s ms/step %.4f accumulate ms/step %.4f' % ('$ablate', d['ms_per_step'], d['kernel_ms_per_step']['accumulate']))"
done
