#!/bin/bash
# Per-kernel times of one bench workload under rocprofv3 --kernel-trace --stats: tools/profile/profile_workload.sh <workload> [tag]
set -u
WL=$1; TAG=${2:-$1}
OUT=$PWD/gpurun_out/kstats_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_w
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_w -o w -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 20 --warmup 2 --repeats 2 --no-cpu > $OUT/stdout.txt 2> $OUT/stderr.txt
echo "rc=$?"
find /tmp/rp_w -name '*kernel_stats.csv' | while read f; do cp "$f" $OUT/kernel_stats.csv; done
python3 - $OUT/kernel_stats.csv <<'PY' | tee $OUT/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print("%-66s calls=%5s avg_us=%8.2f total_us=%10.1f" % (r["Name"][:66], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
