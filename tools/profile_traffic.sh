#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one bench workload (two PMC passes, nothing else): tools/profile_traffic.sh <workload>
set -u
WL=$1
OUT=$PWD/gpurun_out/traffic_$WL
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_$c
  timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d /tmp/rp_$c -o $c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu --workload $WL > $OUT/$c.stdout 2> $OUT/$c.stderr
  echo "$c rc=$?"
  find /tmp/rp_$c -name '*counter_collection.csv' | while read f; do cp "$f" "$OUT/pmc_${c}_counter_collection.csv"; done
done
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT
