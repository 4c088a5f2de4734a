#!/bin/bash
# On the GPU box: the default workload with rings of 1..12 distinct frames (ring 1-2 fit the 256 MiB Infinity Cache).
set -u
mkdir -p gpurun_out
OUT=gpurun_out/ring_sweep.txt
: > $OUT
for ring in ${RINGS:-1 2 3 6 12}; do
  line=$(timeout -k 10 120 python3 bench.py --ring $ring --steps 240 --warmup 24 --repeats 7 --no-cpu "$@" 2>/dev/null | tail -1)
  echo "ring $ring $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("%.2f us (min %.2f max %.2f) frac %.4f" % (r["avg_launch_ms_hip_events"]*1e3, r["avg_launch_ms_min_max"][0]*1e3, r["avg_launch_ms_min_max"][1]*1e3, r["frac"]))' 2>&1 | tail -1)" | tee -a $OUT
done
