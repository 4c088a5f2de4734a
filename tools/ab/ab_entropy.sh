#!/bin/bash
# On the GPU box: per-kernel times of the entropy stage (rocprofv3 --kernel-trace --stats over tools/profile/entropy_profile.py) and the
# planes -> .jpg bench line for every ab/libjpezy_<name>.so.   usage: tools/ab/ab_entropy.sh
set -u
mkdir -p gpurun_out
OUT=$PWD/gpurun_out/ab_entropy.txt
: > $OUT
export TMPDIR=/tmp
ROOT=$PWD
for lib in $ROOT/ab/libjpezy_*.so; do
  name=$(basename $lib .so); name=${name#libjpezy_}
  line=$(JPEZY_LIB=$lib timeout -k 10 200 python3 bench.py --workload encode4096_jpg --steps 100 --warmup 10 --repeats 5 --no-cpu 2>/dev/null | tail -1)
  echo "$name $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("planes->jpg %.2f us/step, entropy stage %.2f us, frac %.4f" % (d["ms_per_step"]*1e3, r["avg_step_ms_hip_events"]*1e3, r["frac"]))' 2>&1 | tail -1)" | tee -a $OUT
  (cd /tmp && rm -rf /tmp/rp_en && JPEZY_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_en -o en -- python3 $ROOT/tools/profile/entropy_profile.py > /dev/null 2>&1
   f=$(find /tmp/rp_en -name '*kernel_stats.csv' | head -1)
   python3 - "$f" "$name" <<'PY' | tee -a $OUT
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "entropy" in r["Name"]:
        print(f"  {sys.argv[2]} {r['Name'][:60]:60s} calls={r['Calls']:>3s} avg_us={float(r['AverageNs'])/1e3:7.2f}")
PY
  )
  if [ "${PMC:-0}" = "1" ]; then
  (cd /tmp && rm -rf /tmp/rp_en2 && JPEZY_LIB=$lib timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/rp_en2 -o en -- python3 $ROOT/tools/profile/entropy_profile.py > /dev/null 2>&1
   f=$(find /tmp/rp_en2 -name '*counter_collection.csv' | head -1)
   python3 - "$f" "$name" <<'PY' | tee -a $OUT
import csv, sys
from collections import defaultdict
acc = defaultdict(float); cnt = defaultdict(int)
for row in csv.DictReader(open(sys.argv[1])):
    if "code_tiles" in row["Kernel_Name"]:
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
w = acc["SQ_WAVES"] / max(cnt["SQ_WAVES"], 1)
print(" ", sys.argv[2], "code_tiles per wave:", {k: round(acc[k] / cnt[k] / w, 1) for k in acc if k != "SQ_WAVES"}, "waves", w)
PY
  )
  fi
done
