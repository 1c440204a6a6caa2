#!/bin/bash
# The engine's HOST code under AddressSanitizer (device code unchanged: -fno-gpu-sanitize; GPU ASan
# is not available on this pool).  Build in the build container:
#   hipcc -O1 -g --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -shared \
#         -fsanitize=address -fno-gpu-sanitize -shared-libsan pylbl_amd/csrc/engine.hip \
#         -o pylbl_amd/liblbl_amd_asan.so -ldl
# then on the GPU box, from the repo root:  scripts/checks/host_asan.sh [pytest args]
set -o pipefail
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
# (the shipped library stays where it is: the engine loads the one $PYLBL_AMD_LIBRARY names)
export PYLBL_AMD_LIBRARY=$(pwd)/pylbl_amd/liblbl_amd_asan.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1:log_path=gpurun_out/asan
LD_PRELOAD=$RT timeout -k 10 900 python -m pytest "${@:-tests/test_gpu_api.py}" -x -q -m gpu -p no:cacheprovider
rc=$?
ls gpurun_out/asan* 2>/dev/null && head -60 gpurun_out/asan*
exit $rc
