#!/bin/bash
# Instruction-mix and issue counters of the default bench command, one rocprofv3 --pmc pass per
# group (kernel trace only, as the pool requires).  Usage on the GPU box, from the repo root:
#   scripts/profile_counters.sh <tag> [bench args]
# EXTRAS=continuum adds the continuum / cross-section legs (bench.py --extras) to the profiled command.
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for group in "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
             "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --extras ${EXTRAS:-none} "$@" > $OUT/bench$i.json 2> $OUT/pass$i.err || exit 1
done
find $OUT -name "*counter_collection.csv"
