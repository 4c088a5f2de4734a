#!/bin/bash
# On the GPU box: FETCH_SIZE / WRITE_SIZE per launch (KB as rocprofv3 reports them; FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md)
# of the kernels of one bench workload whose names match a regex, for every ab/libjpezy_<name>.so or the in-tree library.
#   usage: tools/profile/pmc_traffic.sh <workload> <kernel-regex> [lib names...]
set -u
WL=$1; RX=$2; shift 2
mkdir -p gpurun_out
OUT=$PWD/gpurun_out/pmc_traffic.txt
: > $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
[ $# -eq 0 ] && set -- ""
for name in "$@"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/rp_pt
    if [ -n "$name" ]; then export JPEZY_LIB=$ROOT/ab/libjpezy_$name.so; fi
    timeout -k 10 200 rocprofv3 --pmc $ctr --output-format csv -d /tmp/rp_pt -o pmc -- python3 $ROOT/bench.py --workload $WL --steps 10 --warmup 2 --repeats 1 --no-cpu > /dev/null 2>&1
    f=$(find /tmp/rp_pt -name '*counter_collection.csv' | head -1)
    python3 - "$f" "$RX" "${name:-in-tree}" <<'PY' | tee -a $OUT
import csv, re, sys
from collections import defaultdict
acc = defaultdict(float); cnt = defaultdict(int)
for row in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], row["Kernel_Name"]):
        k = (row["Kernel_Name"][:40], row["Counter_Name"])
        acc[k] += float(row["Counter_Value"]); cnt[k] += 1
for (k, c), v in acc.items():
    kb = v / cnt[(k, c)]
    print(f"{sys.argv[3]:10s} {k:40s} {c:10s} per launch {kb:12.1f} KB" + (f"  -> x2 = {2 * kb / 1e3:8.2f} MB" if c == "FETCH_SIZE" else f"  = {kb / 1e3:8.2f} MB"), f"({cnt[(k, c)]} launches)")
PY
  done
done
