#!/bin/bash
# On the GPU box: single-file GPU Huffman decoder, subsequence size (ab/libjpezy_<name>.so built with -DJPEZY_SUBSEQ_BITS=...) x speculation
# distance (JPEZY_HUFFDEC_OVERFLOW, in subsequences): host-decoder share of the fuzz corpus, launches per file, 4096^2 / 1080p timings.
#   CONFIGS="cur:3 cur:1 s256:12 s256:4" CASES=100 bash tools/ab/ab_huffdec_sweep.sh
set -u
mkdir -p gpurun_out
OUT=gpurun_out/ab_huffdec_sweep.txt
: > $OUT
for cfg in ${CONFIGS}; do
  name=${cfg%%:*}; ovf=${cfg##*:}
  lib=$PWD/ab/libjpezy_$name.so
  echo "== $name overflow $ovf" | tee -a $OUT
  JPEZY_HUFFDEC_OVERFLOW=$ovf JPEZY_LIB=$lib timeout -k 10 300 python3 tools/fuzz/fuzz_huffdec.py ${CASES:-100} 1 2>&1 | grep -v amdgpu.ids | grep -E "files identical|time over|MISMATCH|Error|error" | tee -a $OUT
  JPEZY_HUFFDEC_OVERFLOW=$ovf JPEZY_LIB=$lib timeout -k 10 300 python3 tools/measure/measure_huffdec.py 2>&1 | grep -E "GPU Huffman decode|libjpeg" | tee -a $OUT
done
