#!/bin/bash
# On the GPU box: the shipped library under JPEZY_ENC_VARIANT=1 (one quad per wave) and =2 (persistent workgroups), interleaved
# over ROUNDS rounds of the benchmark's default workload.   usage: [ROUNDS=3] [VARIANTS="1 2"] tools/ab/ab_variants.sh [bench args...]
set -u
mkdir -p gpurun_out
OUT=gpurun_out/ab_variants.txt
: > $OUT
for round in $(seq 1 ${ROUNDS:-3}); do
  for v in ${VARIANTS:-1 2}; do
    line=$(JPEZY_ENC_VARIANT=$v timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --repeats 7 --no-cpu --no-others --no-native-multi "$@" 2>/dev/null | tail -1)
    echo "variant $v $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("%.2f us (min %.2f max %.2f) frac %.4f" % (r["avg_launch_ms_hip_events"]*1e3, r["avg_launch_ms_min_max"][0]*1e3, r["avg_launch_ms_min_max"][1]*1e3, r["frac"]))' 2>&1 | tail -1)" | tee -a $OUT
  done
done
