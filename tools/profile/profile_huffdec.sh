#!/bin/bash
# Per-kernel times of the GPU Huffman decoder (tools/measure/measure_huffdec.py under rocprofv3 --kernel-trace --stats).
set -u
OUT=$PWD/gpurun_out/huffdec_prof
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_hd
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_hd -o hd -- python3 $GRAFT_REPO_ROOT/tools/measure/measure_huffdec.py > $OUT/stdout.txt 2> $OUT/stderr.txt
echo "rc=$?"
find /tmp/rp_hd -name '*kernel_stats.csv' | while read f; do cp "$f" $OUT/kernel_stats.csv; done
head -25 $OUT/kernel_stats.csv | cut -c1-200
