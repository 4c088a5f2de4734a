#!/bin/bash
# Second-level PMC passes (latency / instruction-mix counters) for one bench workload; see tools/profile/profile.sh.
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/prof2_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 30 --warmup 5 --no-cpu $*"
cd /tmp
run() { local name=$1; shift
  rm -rf /tmp/rp_$name
  timeout -k 10 240 rocprofv3 "$@" --output-format csv -d /tmp/rp_$name -o $name -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/$name.stdout 2> $OUT/$name.stderr
  echo "$name rc=$?" >> $OUT/summary.txt
  find /tmp/rp_$name -name '*counter_collection.csv' | while read f; do cp "$f" "$OUT/${name}_$(basename $f)"; done
}
: > $OUT/summary.txt
run pmc_a --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVES
run pmc_b --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
run pmc_c --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_IFETCH SQ_INST_CYCLES_SALU
run pmc_d --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM
python3 $GRAFT_REPO_ROOT/tools/profile/summarize_prof.py $OUT >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
