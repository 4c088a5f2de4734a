#!/bin/bash
# On the GPU box: the GPU Huffman decoder of every ab/libjpezy_<name>.so (tools/ab/ab_build.py, e.g. sub512:-DJPEZY_SUBSEQ_BITS=512)
# through the differential fuzzer (host-decoder share, launches per file), the single-file probe and the 256-file batch.
set -u
mkdir -p gpurun_out
OUT=gpurun_out/ab_huffdec.txt
: > $OUT
for lib in ab/libjpezy_*.so; do
  name=$(basename $lib .so); name=${name#libjpezy_}
  echo "== $name" | tee -a $OUT
  JPEZY_LIB=$PWD/$lib timeout -k 10 400 python3 tools/fuzz/fuzz_huffdec.py ${CASES:-150} 1 2>&1 | grep -v amdgpu.ids | tee -a $OUT
  JPEZY_LIB=$PWD/$lib timeout -k 10 300 python3 tools/measure/measure_huffdec.py 2>&1 | grep -E "GPU Huffman decode|jpezy_decode_jpeg_batch|libjpeg" | tee -a $OUT
  JPEZY_LIB=$PWD/$lib timeout -k 10 300 python3 tools/measure/measure_decode_batch_raw.py $PWD/$lib 2>&1 | grep -E "^libjpezy" | tee -a $OUT
done
