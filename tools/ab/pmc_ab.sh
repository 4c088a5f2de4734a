#!/bin/bash
# On the GPU box: SQ counters of the timed encode / decode kernel for every ab/libjpezy_<name>.so named in LIBS (default: all), one
# rocprofv3 --pmc pass per counter group of at most 8, totals PER LAUNCH (the persistent encode kernel has few, long waves: per-wave
# figures do not compare with the one-quad kernel's).   usage: [LIBS="base ps_14_1"] tools/ab/pmc_ab.sh [bench args...]
set -u
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out/pmc_ab.txt
: > $OUT
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
G2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC"
cd /tmp
for name in ${LIBS:-$(cd $ROOT/ab; ls libjpezy_*.so | sed 's/libjpezy_//; s/\.so//')}; do
  lib=$ROOT/ab/libjpezy_$name.so
  for grp in "$G1" "$G2" "$G3"; do
    rm -rf /tmp/rp_ab
    JPEZY_ALLOW_EXPERIMENT=1 JPEZY_LIB=$lib timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/rp_ab -o pmc -- python3 $ROOT/bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu --no-others --no-native-multi "$@" > /tmp/rp_ab.log 2>&1
    f=$(find /tmp/rp_ab -name '*counter_collection.csv' | head -1)
    if [ -z "$f" ]; then echo "$name: no counters for [$grp]" | tee -a $OUT; tail -3 /tmp/rp_ab.log | tee -a $OUT; continue; fi
    python3 - "$f" "$name" <<'PY' | tee -a $OUT
import csv, sys
from collections import defaultdict
acc = defaultdict(float); cnt = defaultdict(int)
for row in csv.DictReader(open(sys.argv[1])):
    if "jpezy" in row["Kernel_Name"] and ("fdct_quant" in row["Kernel_Name"] or "dequant_idct" in row["Kernel_Name"]):
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
print(sys.argv[2], "per launch:", {k: round(acc[k] / cnt[k]) for k in acc})
PY
  done
done
