#!/bin/bash
# Every profile the round's record cites, in one go (GPU box, from the repo root):
#   scripts/profile_round.sh <tag>        e.g. r03a
# default workload with the slot kernels (stats + FETCH_SIZE + WRITE_SIZE passes), its instruction
# counters, the --pedestal leg, BASELINE configs[1], and 8 standard-atmosphere levels with the
# pedestal.  Summaries: scripts/summarize_profile.py / summarize_counters.py (run in the build
# container on the merged gpurun_out/).
set -o pipefail
TAG=${1:-r03a}
EXTRAS=continuum scripts/profile_bench.sh ${TAG} > gpurun_out/profile_${TAG}.log 2>&1 || exit 1
echo "default done"
EXTRAS=continuum scripts/profile_counters.sh ${TAG} >> gpurun_out/profile_${TAG}.log 2>&1 || exit 1
echo "counters done"
scripts/profile_bench.sh ${TAG}_pedestal --pedestal >> gpurun_out/profile_${TAG}.log 2>&1 || exit 1
echo "pedestal done"
scripts/profile_bench.sh ${TAG}_config1 --config 1 >> gpurun_out/profile_${TAG}.log 2>&1 || exit 1
echo "config1 done"
scripts/profile_bench.sh ${TAG}_standard8 --levels-per-gpu 8 --profile standard --pedestal >> gpurun_out/profile_${TAG}.log 2>&1 || exit 1
echo "standard8 done"
scripts/profile_bench.sh ${TAG}_farfield --farfield --pedestal >> gpurun_out/profile_${TAG}.log 2>&1 || exit 1
echo "farfield done"
# round 5: the far-field legs' own counters (what farfield_option.*.roofline reads)
scripts/profile_bench.sh ${TAG}_farfield_plain --farfield >> gpurun_out/profile_${TAG}.log 2>&1 || exit 1
scripts/profile_counters.sh ${TAG}_farfield --farfield --pedestal >> gpurun_out/profile_${TAG}.log 2>&1 || exit 1
scripts/profile_counters.sh ${TAG}_farfield_plain --farfield >> gpurun_out/profile_${TAG}.log 2>&1 || exit 1
echo "farfield counters done"
