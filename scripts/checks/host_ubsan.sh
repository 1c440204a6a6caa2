#!/bin/bash
# The engine's HOST code under UndefinedBehaviorSanitizer (device code unchanged).  Build in the
# build container:
#   hipcc -O1 -g --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -shared \
#         -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize=vptr,function -shared-libsan \
#         pylbl_amd/csrc/engine.hip -o pylbl_amd/liblbl_amd_ubsan.so -ldl
# then on the GPU box, from the repo root:  scripts/checks/host_ubsan.sh [pytest args]
set -o pipefail
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
# (the shipped library stays where it is: the engine loads the one $PYLBL_AMD_LIBRARY names)
export PYLBL_AMD_LIBRARY=$(pwd)/pylbl_amd/liblbl_amd_ubsan.so
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=gpurun_out/ubsan
LD_PRELOAD=$RT timeout -k 10 900 python -m pytest "${@:-tests/test_gpu_api.py}" -x -q -m gpu -p no:cacheprovider
rc=$?
ls gpurun_out/ubsan* 2>/dev/null && head -80 gpurun_out/ubsan*
exit $rc
