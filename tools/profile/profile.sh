#!/bin/bash
# Collect the rocprofv3 evidence for one bench workload on the GPU box (run via gpurun from the repo root):
#   tools/profile/profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_*.csv,summary.txt}; copy summaries into profiles/.
# Counters are collected in their own passes (no tracing flags together with --pmc).
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 40 --warmup 5 --repeats 2 --no-cpu --no-others --no-native-multi $*"   # (--no-others: the default workload would otherwise append configs[2] and [4], whose kernels share truncated names with the timed one)
cd /tmp
run() { # name, rocprof flags...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  timeout -k 10 240 rocprofv3 "$@" --output-format csv -d /tmp/rp_$name -o $name -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/$name.stdout 2> $OUT/$name.stderr
  echo "$name rc=$?" >> $OUT/summary.txt
  find /tmp/rp_$name -name '*.csv' | while read f; do cp "$f" "$OUT/${name}_$(basename $f)"; done
}
: > $OUT/summary.txt
run kt --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run pmc_sq2 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
run pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
python3 $GRAFT_REPO_ROOT/tools/profile/summarize_prof.py $OUT >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
