#!/bin/bash
# rocprofv3 --kernel-trace --stats of EXACTLY the driver's command (python3 bench.py --gpus 1 --steps 20 --warmup 5): the per-kernel table whose
# average for the timed kernel must agree with the line's roofline.avg_launch_ms_hip_events.   usage: tools/profile/driver_command_trace.sh <round>
set -u
R=${1:-r05}
OUT=$PWD/gpurun_out/driver_$R
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_drv
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_drv -o drv -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${R}_bench_under_rocprof_driver_command.json 2> $OUT/stderr.txt
echo "rc=$?"
f=$(find /tmp/rp_drv -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/${R}_driver_command_kernel_stats.csv
python3 - "$f" <<'PY' | tee $OUT/${R}_driver_command_summary.txt
import csv, sys
print("rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5")
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-100s calls=%5s avg_us=%8.2f min_us=%8.2f max_us=%8.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
