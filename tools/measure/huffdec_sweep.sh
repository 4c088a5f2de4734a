#!/bin/bash
# On the GPU box: GPU Huffman decoder time for every ab/libjpezy_s<bits>.so (subsequence size) x speculation distance.
set -u
mkdir -p gpurun_out
OUT=gpurun_out/huffdec_sweep.txt
: > $OUT
for lib in ab/libjpezy_s*.so; do
  name=$(basename $lib .so)
  for ov in ${OVS:-1 2 3 4 6 8 12}; do
    r=$(JPEZY_LIB=$PWD/$lib JPEZY_HUFFDEC_OVERFLOW=$ov timeout -k 10 120 python3 tools/measure/measure_huffdec.py 2>/dev/null | grep "GPU Huffman decode" | head -4 | sed 's/.*GPU Huffman decode \([0-9.]* ms ([0-9]* passes)\).*identical: \(.*\)/\1 \2/' | tr '\n' '|')
    echo "$name overflow=$ov : $r" | tee -a $OUT
  done
done
