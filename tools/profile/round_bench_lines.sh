#!/bin/bash
# The bench lines of a round on the GPU box (run from the repo root): the driver's command, the default line of every workload,
# the N = 1 batch pipeline.  Written under gpurun_out/bench_<round>/; copied to profiles/ afterwards.   usage: tools/profile/round_bench_lines.sh r04
set -u
R=${1:-r04}
OUT=gpurun_out/bench_$R
mkdir -p $OUT
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${R}_bench_k20.json
echo "k20 done"
for w in encode4096 decode4096 gray8k gray8k_decode batch1080p encode4096_jpg decode4096_jpg; do
  python3 bench.py --workload $w 2>$OUT/${R}_bench_$w.err | tail -1 > $OUT/${R}_bench_$w.json
  echo "$w done"
done
python3 bench.py --batch --no-cpu 2>/dev/null | tail -1 > $OUT/${R}_bench_batch_n1.json
echo finished
