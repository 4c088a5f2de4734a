#!/bin/bash
# On the GPU box: the encode kernel variants of ONE library benched interleaved (bench.py --variant V), ROUNDS rounds, then the
# VALU / MFMA instruction counts per wave of each from a --pmc pass.   usage: [ROUNDS=3] tools/ab_variants.sh "1 2" [bench args]
set -u
VARS=$1; shift
mkdir -p gpurun_out
OUT=gpurun_out/ab_variants.txt
: > $OUT
for round in $(seq 1 ${ROUNDS:-3}); do
  for v in $VARS; do
    line=$(timeout -k 10 120 python3 bench.py --variant $v --steps 200 --warmup 20 --repeats 7 --no-cpu "$@" 2>/dev/null | tail -1)
    echo "variant $v $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("%.2f us (min %.2f max %.2f) frac %.4f  fallbacks/step %.0f" % (r["avg_launch_ms_hip_events"]*1e3, r["avg_launch_ms_min_max"][0]*1e3, r["avg_launch_ms_min_max"][1]*1e3, r["frac"], d["exact_fallbacks_per_step"]))' 2>&1 | tail -1)" | tee -a $OUT
  done
done
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for v in $VARS; do
  rm -rf /tmp/rp_abv
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/rp_abv -o pmc -- python3 $ROOT/bench.py --variant $v --steps 20 --warmup 2 --repeats 1 --no-cpu "$@" > /dev/null 2>&1
  f=$(find /tmp/rp_abv -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$v" <<'PY' | tee -a $ROOT/$OUT
import csv, sys
from collections import defaultdict
acc = defaultdict(float); cnt = defaultdict(int)
for row in csv.DictReader(open(sys.argv[1])):
    if "fdct_quant" in row["Kernel_Name"]:
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
w = acc["SQ_WAVES"] / max(cnt["SQ_WAVES"], 1)
print("variant", sys.argv[2], "per wave:", {k: round(acc[k] / cnt[k] / w, 1) for k in acc if k != "SQ_WAVES"}, "waves", w)
PY
done
