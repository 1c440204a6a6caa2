#!/bin/bash
# The engine's HOST code under ThreadSanitizer (device code unchanged): the tests that call one
# engine from several threads.  Build in the build container:
#   hipcc -O1 -g --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -shared \
#         -Xarch_host -fsanitize=thread -shared-libsan pylbl_amd/csrc/engine.hip \
#         -o pylbl_amd/liblbl_amd_tsan.so -ldl
# then on the GPU box, from the repo root:  scripts/checks/host_tsan.sh [pytest args]
# (Only accesses made by the engine's own code are seen; the interpreter and the HIP runtime are
# not instrumented, their locks are -- pthread calls are intercepted.)
set -o pipefail
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
# (the shipped library stays where it is: the engine loads the one $PYLBL_AMD_LIBRARY names)
export PYLBL_AMD_LIBRARY=$(pwd)/pylbl_amd/liblbl_amd_tsan.so
export TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:log_path=gpurun_out/tsan
LD_PRELOAD=$RT timeout -k 10 600 python -m pytest "${@:-tests/test_gpu_threads.py}" -x -q -m gpu -p no:cacheprovider
rc=$?
ls gpurun_out/tsan* 2>/dev/null && grep -h "WARNING\|SUMMARY" gpurun_out/tsan* | sort | uniq -c | sort -rn | head -30
exit $rc
