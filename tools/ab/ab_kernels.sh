#!/bin/bash
# On the GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) of one bench workload for every ab/libjpezy_<name>.so
# (tools/ab/ab_build.py).   usage: tools/ab/ab_kernels.sh <workload> [kernel-name filter]
set -u
WL=$1; FILTER=${2:-.}
mkdir -p gpurun_out
OUT=$PWD/gpurun_out/ab_kernels.txt
: > $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for lib in $ROOT/ab/libjpezy_*.so; do
  name=$(basename $lib .so); name=${name#libjpezy_}
  rm -rf /tmp/rp_abk
  JPEZY_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_abk -o w -- python3 $ROOT/bench.py --workload $WL --steps 20 --warmup 2 --repeats 2 --no-cpu --no-others --no-native-multi > /dev/null 2>&1
  f=$(find /tmp/rp_abk -name '*kernel_stats.csv' | head -1)
  echo "== $name" | tee -a $OUT
  python3 - "$f" "$FILTER" <<'PY' | tee -a $OUT
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    if re.search(sys.argv[2], r["Name"]):
        print("  %-60s calls=%5s avg_us=%8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
