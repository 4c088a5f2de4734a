#!/bin/bash
# Refresh the evidence of a round on the GPU box: tools/profile/final_profiles.sh <tag>   (writes under gpurun_out/final_<tag>/)
set -u
TAG=${1:-r01j}
OUT=$PWD/gpurun_out/final_$TAG
mkdir -p $OUT
bash tools/profile/profile.sh ${TAG}_enc --workload encode4096 > $OUT/profile_enc.log 2>&1
cp gpurun_out/prof_${TAG}_enc/summary.txt $OUT/${TAG}_summary.txt
cp gpurun_out/prof_${TAG}_enc/kt_*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null
tail -1 gpurun_out/prof_${TAG}_enc/kt.stdout > $OUT/${TAG}_bench_under_rocprof_encode4096.json
bash tools/profile/profile.sh ${TAG}_dec --workload decode4096 > $OUT/profile_dec.log 2>&1
cp gpurun_out/prof_${TAG}_dec/summary.txt $OUT/${TAG}_dec_summary.txt
cp gpurun_out/prof_${TAG}_dec/kt_*kernel_stats.csv $OUT/${TAG}_dec_kernel_stats.csv 2>/dev/null
tail -1 gpurun_out/prof_${TAG}_dec/kt.stdout > $OUT/${TAG}_bench_under_rocprof_decode4096.json
for w in encode4096 decode4096 gray8k batch1080p gray8k_decode; do
  python3 bench.py --workload $w 2>/dev/null | tail -1 > $OUT/${TAG}_bench_$w.json
  python3 bench.py --workload $w --pipelined --no-cpu 2>/dev/null | tail -1 > $OUT/${TAG}_bench_pipelined_$w.json
  echo "$w done"
done
python3 tools/measure/measure_huffdec.py > $OUT/${TAG}_huffdec.txt 2>&1
python3 tools/measure/measure_entropy.py > $OUT/${TAG}_entropy.txt 2>&1
echo finished
