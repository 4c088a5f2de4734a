#!/bin/bash
# On the GPU box: PMC counters (one rocprofv3 --pmc pass per group, no tracing beside it) of the kernels of one bench workload
# whose names match a regex.   usage: tools/profile/pmc_kernel.sh <workload> <kernel-regex> ["CTR1 CTR2 ..." ["CTR..."]]
set -u
WL=$1; RX=$2; shift 2
[ $# -eq 0 ] && set -- "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out/pmc_kernel.txt
: > $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for grp in "$@"; do
  rm -rf /tmp/rp_pk
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/rp_pk -o pmc -- python3 $ROOT/bench.py --workload $WL --steps 10 --warmup 2 --repeats 1 --no-cpu > /dev/null 2>&1
  f=$(find /tmp/rp_pk -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$RX" <<'PY' | tee -a $OUT
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if re.search(sys.argv[2], k):
        k = k[:48]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in acc:
    w = acc[k]["SQ_WAVES"] / max(cnt[k]["SQ_WAVES"], 1)
    print(k, "waves", w, "per wave:", {c: round(acc[k][c] / cnt[k][c] / w, 1) for c in acc[k] if c != "SQ_WAVES"})
PY
done
