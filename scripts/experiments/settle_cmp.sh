#!/bin/bash
# Durations of run_chain_kernel (which, behind a relaxation that settled, only reads the flags and leaves)
# beside the direct accumulate kernel, per build of the library: bash scripts/experiments/settle_cmp.sh <name> ...
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && cd $R
cp pylbl_amd/liblbl_amd.so /tmp/orig.so
for name in "$@"; do
  cp build/liblbl_$name.so pylbl_amd/liblbl_amd.so
  rm -rf gpurun_out/settles_$name
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/settles_$name -- python3 scripts/experiments/chain_settles.py 100 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/settles_$name/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
chain = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3 for r in rows if "run_chain" in r["Kernel_Name"])
print("$name: run_chain launches", len(chain), "median %.1f  p90 %.1f  over 100 us: %d  over 1000 us: %d  max %.1f us" % (chain[len(chain)//2], chain[int(len(chain)*0.9)], sum(1 for x in chain if x > 100), sum(1 for x in chain if x > 1000), chain[-1]))
PY
done
cp /tmp/orig.so pylbl_amd/liblbl_amd.so
