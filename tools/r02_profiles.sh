#!/bin/bash
# round-2 evidence: rocprofv3 summaries of the default command's kernel (encode) and of the decode kernel, PMC traffic of
# every workload, bench lines of every workload.  Run via gpurun from the repo root.
set -u
mkdir -p gpurun_out/r02p
bash tools/profile.sh r02_enc > /dev/null 2>&1; cp gpurun_out/prof_r02_enc/summary.txt gpurun_out/r02p/r02_summary.txt; echo "enc prof done"
bash tools/profile.sh r02_dec --workload decode4096 > /dev/null 2>&1; cp gpurun_out/prof_r02_dec/summary.txt gpurun_out/r02p/r02_dec_summary.txt; echo "dec prof done"
for wl in gray8k batch1080p gray8k_decode; do
  bash tools/profile_traffic.sh $wl > gpurun_out/r02p/r02_traffic_$wl.txt 2>&1; echo "traffic $wl done"
done
for wl in encode4096 decode4096 gray8k batch1080p gray8k_decode encode4096_jpg; do
  timeout -k 10 300 python3 bench.py --workload $wl --pipelined > gpurun_out/r02p/bench_$wl.json 2> gpurun_out/r02p/bench_$wl.err; echo "bench $wl rc=$?"
done
