#!/bin/bash
# On the GPU box: timeline of ONE jpezy_read_jpeg_gpu call on a 4096x4096 random-pixel file (tools/measure/huffdec_one.py) -- every kernel and
# memory copy with its start offset, duration and the gap in front of it.  JPEZY_LIB / JPEZY_HUFFDEC_OVERFLOW select the build / knob.
#   bash tools/profile/huffdec_timeline.sh [W H] > gpurun_out/huffdec_timeline.txt
set -u
export TMPDIR=/tmp
ROOT=$PWD
W=${1:-4096}; H=${2:-4096}
cd /tmp
rm -rf /tmp/rp_hl
timeout -k 10 280 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/rp_hl -o hl -- python3 $ROOT/tools/measure/huffdec_one.py $W $H 2> /tmp/rp_hl.err | grep -v amdgpu.ids
python3 - <<'PY'
import csv, glob
k = sorted(glob.glob("/tmp/rp_hl/**/*kernel_trace.csv", recursive=True))
m = sorted(glob.glob("/tmp/rp_hl/**/*memory_copy_trace.csv", recursive=True))
ev = []
for r in csv.DictReader(open(k[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("jpezy_dev::", "")[:48]))
if m:
    for r in csv.DictReader(open(m[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
starts = [i for i, e in enumerate(ev) if "unstuff_count" in e[2]]
# a call = from the scan upload in front of an unstuff_count launch to the event in front of the next call's upload
def call_range(j):
    a = starts[j]
    while a > 0 and ev[a - 1][0] > ev[starts[j]][0] - 400_000 and (j == 0 or a - 1 > starts[j - 1]) and ("copy" in ev[a - 1][2] or "fill" in ev[a - 1][2].lower()):
        a -= 1
    return a
j = len(starts) - 3
a, b = call_range(j), call_range(j + 1)
t0, prev = ev[a][0], ev[a][0]
tot = 0
print(f"call {j}: {b - a} events, span {(ev[b - 1][1] - t0) / 1e3:.1f} us; period to the next call's first event {(ev[b][0] - t0) / 1e3:.1f} us")
for s, e, n in ev[a:b]:
    print(f"  +{(s - t0) / 1e3:8.1f} us  gap {(s - prev) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  {n}")
    prev = max(prev, e); tot += e - s
print(f"  sum of durations {tot / 1e3:.1f} us")
PY
