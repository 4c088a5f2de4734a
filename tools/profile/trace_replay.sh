#!/bin/bash
# Where does a short replay's time go?  rocprofv3 --kernel-trace --hip-runtime-trace of `bench.py --steps 20` (no counters),
# then per replay: host time of hipGraphLaunch, gap between its start and the first kernel, kernel span, gap to the sync's return.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rm -rf /tmp/rp_tr
timeout -k 10 200 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d /tmp/rp_tr -o tr -- python3 $ROOT/bench.py --steps ${STEPS:-20} --warmup 5 --repeats 4 --no-cpu "$@" > /dev/null 2> /tmp/rp_tr.err
echo "rc=$?"
ls /tmp/rp_tr/*/ | head
python3 - <<'PY' | tee $ROOT/gpurun_out/trace_replay.txt
import csv, glob
k = sorted(glob.glob("/tmp/rp_tr/**/*kernel_trace.csv", recursive=True))
a = sorted(glob.glob("/tmp/rp_tr/**/*hip_api_trace.csv", recursive=True))
print(k, a)
if not (k and a):
    raise SystemExit
K = [r for r in csv.DictReader(open(k[0])) if "fdct_quant" in r["Kernel_Name"]]
A = list(csv.DictReader(open(a[0])))
print("api columns", list(A[0].keys()))
launches = [r for r in A if r["Function"] == "hipGraphLaunch"]
syncs = [r for r in A if r["Function"] in ("hipDeviceSynchronize", "hipStreamSynchronize")]
print(len(K), "kernels", len(launches), "graph launches", len(syncs), "syncs")
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in K)
for L in launches[:6]:
    t0, t1 = int(L["Start_Timestamp"]), int(L["End_Timestamp"])
    inside = [x for x in ks if x[0] >= t0 and x[0] < t0 + 3_000_000]
    nxt = [int(s["End_Timestamp"]) for s in syncs if int(s["End_Timestamp"]) > t1]
    if not inside:
        continue
    n = 0
    # kernels of this replay: consecutive ones from the first after t0 until a gap > 50 us
    run = [inside[0]]
    for x in inside[1:]:
        if x[0] - run[-1][1] > 50_000:
            break
        run.append(x)
    print("hipGraphLaunch host %.1f us | launch start -> first kernel %.1f us | %d kernels span %.1f us (sum %.1f) | last kernel end -> sync return %.1f us" % (
        (t1 - t0) / 1e3, (run[0][0] - t0) / 1e3, len(run), (run[-1][1] - run[0][0]) / 1e3, sum(b - a for a, b in run) / 1e3,
        ((min(nxt) - run[-1][1]) / 1e3) if nxt else -1))
PY
