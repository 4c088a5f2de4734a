#!/bin/bash
# On the GPU box: bench every ab/libjpezy_<name>.so (see tools/ab/ab_build.py), interleaved over ROUNDS rounds, and (PMC=1)
# count the VALU instructions per wave of the timed kernel.   usage: [ROUNDS=3] [PMC=1] tools/ab/ab_run.sh [bench args...]
set -u
mkdir -p gpurun_out
OUT=gpurun_out/ab.txt
: > $OUT
for round in $(seq 1 ${ROUNDS:-3}); do
  for lib in ab/libjpezy_*.so; do
    name=$(basename $lib .so); name=${name#libjpezy_}
    line=$(JPEZY_ALLOW_EXPERIMENT=1 JPEZY_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --repeats 7 --no-cpu --no-others --no-native-multi "$@" 2>/dev/null | tail -1)
    echo "$name $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("%.2f us (min %.2f max %.2f) frac %.4f" % (r["avg_launch_ms_hip_events"]*1e3, r["avg_launch_ms_min_max"][0]*1e3, r["avg_launch_ms_min_max"][1]*1e3, r["frac"]))' 2>&1 | tail -1)" | tee -a $OUT
  done
done
if [ "${PMC:-0}" = "1" ]; then
  export TMPDIR=/tmp
  ROOT=$PWD
  cd /tmp
  for lib in $ROOT/ab/libjpezy_*.so; do
    name=$(basename $lib .so); name=${name#libjpezy_}
    rm -rf /tmp/rp_ab
    JPEZY_ALLOW_EXPERIMENT=1 JPEZY_LIB=$lib timeout -k 10 200 rocprofv3 --pmc ${PMC_LIST:-SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS} --output-format csv -d /tmp/rp_ab -o pmc -- python3 $ROOT/bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu --no-others --no-native-multi "$@" > /dev/null 2>&1
    f=$(find /tmp/rp_ab -name '*counter_collection.csv' | head -1)
    python3 - "$f" "$name" <<'PY' | tee -a $ROOT/$OUT
import csv, sys
from collections import defaultdict
acc = defaultdict(float); cnt = defaultdict(int)
for row in csv.DictReader(open(sys.argv[1])):
    if "jpezy" in row["Kernel_Name"] and ("fdct_quant" in row["Kernel_Name"] or "dequant_idct" in row["Kernel_Name"]):
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
w = acc["SQ_WAVES"] / max(cnt["SQ_WAVES"], 1)
print(sys.argv[2], "per wave:", {k: round(acc[k] / cnt[k] / w, 1) for k in acc if k != "SQ_WAVES"}, "waves", w)
PY
  done
fi
