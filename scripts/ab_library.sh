#!/bin/bash
# Interleaved A/B of two builds of the engine library on one bench command (GPU box, from the repo
# root): scripts/ab_library.sh <out> <variant .so>   with BENCH_ARGS="..." as for ab_farped.sh.
# A = pylbl_amd/liblbl_amd.so as it travelled, B = the variant (built in the build container with
# another -D...: hipcc ... -DLBL_X=0 -o pylbl_amd/liblbl_amd_variant.so).
OUT=$1; VARIANT=$2
: > $OUT
for round in 1 2 3; do
  for which in main variant; do
    if [ $which = main ]; then unset PYLBL_AMD_LIBRARY; else export PYLBL_AMD_LIBRARY=$(realpath $VARIANT); fi
    line=$(python bench.py ${BENCH_ARGS:- } --no-extras --no-cpu-baseline --steps ${STEPS:-40} --warmup 6 2>/dev/null | tail -1)
    ms=$(echo "$line" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms/step  acc_alone %.4f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))")
    echo "round $round  [$which]  ${BENCH_ARGS:-default}: $ms" | tee -a $OUT
  done
done
unset PYLBL_AMD_LIBRARY
