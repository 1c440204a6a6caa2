cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for name in r03c HEAD; do
  dir=$R/build/tree_$name; [ "$name" = HEAD ] && dir=$R
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ffprof_$name -- python3 $dir/bench.py --steps 10 --warmup 3 --no-extras --farfield > /dev/null 2>&1
  f=$(find $R/gpurun_out/ffprof_$name -name "*kernel_stats.csv" | head -1)
  echo "== $name"; head -8 $f | cut -c1-200
done
