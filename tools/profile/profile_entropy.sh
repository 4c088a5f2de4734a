#!/bin/bash
# Per-kernel times of the GPU entropy coder (tools/profile/entropy_profile.py: six 4096x4096 frames) under rocprofv3.
set -u
OUT=$PWD/gpurun_out/entropy_prof
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_en
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_en -o en -- python3 $GRAFT_REPO_ROOT/tools/profile/entropy_profile.py > $OUT/stdout.txt 2> $OUT/stderr.txt
echo "rc=$?"
find /tmp/rp_en -name '*kernel_stats.csv' | while read f; do cp "$f" $OUT/kernel_stats.csv; done
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.1f} total_us={float(r['TotalDurationNs'])/1e3:9.1f}")
PY
