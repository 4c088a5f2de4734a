#!/bin/bash
# On the GPU box: bench every ab/libjpezy_<name>.so (see tools/ab_build.py).  usage: tools/ab_run.sh [bench args...]
set -u
mkdir -p gpurun_out
OUT=gpurun_out/ab.txt
: > $OUT
for round in 1 2; do
  for lib in ab/libjpezy_*.so; do
    name=$(basename $lib .so); name=${name#libjpezy_}
    line=$(JPEZY_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu "$@" 2>/dev/null | tail -1)
    echo "$name $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline().split(" ",0)[0]); print(d["ms_per_step"]*1000, "us  frac", d["roofline"]["frac"])' 2>&1 | tail -1)" | tee -a $OUT
  done
done
