#!/bin/bash
# The un-profiled bench lines the round's record cites (GPU box, from the repo root):
#   scripts/bench_round.sh <tag>     ->  gpurun_out/bench_<tag>*.json (the short lines bench.py prints)
#                                        gpurun_out/bench_<tag>*_full.json (the whole records)
TAG=${1:-r03a}
O=gpurun_out
python bench.py --steps 20 --warmup 5 --full-record $O/bench_${TAG}_full.json > $O/bench_${TAG}.json 2> $O/bench_${TAG}.err || exit 1
echo "default done"
python bench.py --steps 20 --warmup 5 --pedestal --no-extras --full-record $O/bench_${TAG}_pedestal_full.json > $O/bench_${TAG}_pedestal.json 2>> $O/bench_${TAG}.err || exit 1
for c in 0 1 2; do
  python bench.py --steps 20 --warmup 5 --config $c --no-extras --full-record $O/bench_${TAG}_config${c}_full.json > $O/bench_${TAG}_config$c.json 2>> $O/bench_${TAG}.err || exit 1
done
echo "configs 0-2 done"
python bench.py --steps 5 --warmup 2 --config 3 --levels-per-gpu 16 --profile standard --pedestal --no-extras --full-record $O/bench_${TAG}_config3_16levels_full.json > $O/bench_${TAG}_config3_16levels.json 2>> $O/bench_${TAG}.err || exit 1
python bench.py --steps 5 --warmup 2 --config 3 --levels-per-gpu 8 --profile standard --pedestal --no-extras --full-record $O/bench_${TAG}_config3_share8_full.json > $O/bench_${TAG}_config3_share8.json 2>> $O/bench_${TAG}.err || exit 1
python bench.py --steps 5 --warmup 2 --config 4 --levels-per-gpu 4 --profile standard --pedestal --no-extras --full-record $O/bench_${TAG}_config4_4levels_full.json > $O/bench_${TAG}_config4_4levels.json 2>> $O/bench_${TAG}.err || exit 1
python bench.py --steps 5 --warmup 2 --config 4 --levels-per-gpu 4 --profile standard --pedestal --output total --no-extras --full-record $O/bench_${TAG}_config4_4levels_total_full.json > $O/bench_${TAG}_config4_4levels_total.json 2>> $O/bench_${TAG}.err || exit 1
echo "shapes done"
python bench.py --steps 3 --warmup 1 --config 4 --levels-per-gpu 32 --profile standard --pedestal --output total --no-extras --full-record $O/bench_${TAG}_config4_share32_total_full.json > $O/bench_${TAG}_config4_share32_total.json 2>> $O/bench_${TAG}.err || exit 1
python bench.py --steps 3 --warmup 1 --config 4 --levels-per-gpu 32 --profile standard --pedestal --no-extras --full-record $O/bench_${TAG}_config4_share32_full.json > $O/bench_${TAG}_config4_share32.json 2>> $O/bench_${TAG}.err || exit 1
echo "shares done"
python bench.py --steps 10 --warmup 3 --host-output --no-extras --full-record $O/bench_${TAG}_host_output_full.json > $O/bench_${TAG}_host_output.json 2>> $O/bench_${TAG}.err || exit 1
MASTER_PORT=29517 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --full-record $O/bench_${TAG}_gloo2_full.json > $O/bench_${TAG}_gloo2.json 2>> $O/bench_${TAG}.err || exit 1
# the bare form (bench.py starts its two ranks itself), kernels and exchange ordered on the device
PYLBL_AMD_ORDER_ON_DEVICE=1 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --full-record $O/bench_${TAG}_gloo2_bare_full.json > $O/bench_${TAG}_gloo2_bare.json 2>> $O/bench_${TAG}.err || exit 1
python bench.py --steps 20 --warmup 5 --farfield --pedestal --no-extras --full-record $O/bench_${TAG}_farfield_pedestal_full.json > $O/bench_${TAG}_farfield_pedestal.json 2>> $O/bench_${TAG}.err || exit 1
echo "all done"
