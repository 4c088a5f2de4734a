#!/bin/bash
# The scaling measurement of BASELINE configs[3] on a node with SEVERAL MI355X -- written for the day such a node is offered.
#
#   tools/measure/scale_node.sh [N_MAX=8] [OUT=gpurun_out/scale_node.jsonl]
#
# NOTHING HERE HAS EVER RUN ON MORE THAN ONE GPU: the builder's boxes have one, the driver's 8-GPU tier was skipped in rounds 1-5.
# On a one-GPU box the script starts, runs its N = 1 legs and stops; that is all that could be verified.
# What is sharded: the caller's loop over independent encoder objects (ref encoder/jpezy_encoder.hpp:38-77; inside a frame the MCU
# loop :55-67); frames share nothing, so the only cross-device traffic is the gather of the results.
#
# In order, every leg in a FRESH process, one JSON line each (appended to OUT):
#   1. harness, one process per GPU over RCCL:  python bench.py --gpus N   for N = 1, 2, 4, 8 (those <= the GPUs present)
#        the line's "batch" object carries the split SURVEY.md 8(e) asks for: kernel_only / gather / end_to_end / end_to_end_jpg
#   2. native entry, ONE host process (jpezy_multi_create / jpezy_multi_encode; gather by hipMemcpyPeerAsync) on devices 0..N-1:
#        on_root_device = 0  every lane delivers to host memory over its own PCIe link
#        on_root_device = 1  results gathered into devices[0]'s memory over xGMI
#      per leg: end-to-end wall time, per-lane kernel time (HIP events) and wall time, bytes up / brought back, and a check of two
#      frames against the single-frame entry point
set -u
NMAX=${1:-8}
OUT=${2:-gpurun_out/scale_node.jsonl}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python3 -c 'import torch; print(torch.cuda.device_count())')
echo "scale_node: $HAVE GPU(s) visible, N up to $NMAX" >&2
for N in 1 2 4 8; do
  [ "$N" -le "$NMAX" ] && [ "$N" -le "$HAVE" ] || continue
  echo "== harness (RCCL), N = $N" >&2
  timeout -k 10 1500 python3 bench.py --gpus "$N" --steps 20 --warmup 5 --no-cpu --no-native-multi --no-others $([ "$N" = 1 ] && echo --batch) \
    | tail -1 | python3 -c 'import sys, json; d = json.loads(sys.stdin.readline()); d["leg"] = "harness_rccl"; print(json.dumps(d))' >> "$OUT" \
    || echo "{\"leg\": \"harness_rccl\", \"n_gpus\": $N, \"error\": \"bench.py failed\"}" >> "$OUT"
done
for N in 1 2 4 8; do
  [ "$N" -le "$NMAX" ] && [ "$N" -le "$HAVE" ] || continue
  for ROOTDEV in 0 1; do
    echo "== native entry, N = $N, on_root_device = $ROOTDEV" >&2
    timeout -k 10 900 python3 tools/measure/native_multi_leg.py --gpus "$N" --on-root-device "$ROOTDEV" >> "$OUT" \
      || echo "{\"leg\": \"native_multi\", \"n_gpus\": $N, \"on_root_device\": $ROOTDEV, \"error\": \"leg failed\"}" >> "$OUT"
  done
done
echo "scale_node: $(wc -l < "$OUT") line(s) in $OUT" >&2
