#!/bin/bash
# CLI end to end on a 4096x4096 P3 file (GPU box): parser threads 1 vs all cores, encode, decode
set -u
cd /tmp
python3 - <<'PY'
import numpy as np
W=H=4096
a=np.random.default_rng(0).integers(0,256,(W*H,3),dtype=np.uint8)
with open('/tmp/big.ppm','w') as f:
    f.write(f"P3\n{W} {H}\n255\n"); np.savetxt(f,a,fmt='%d')
PY
B=$GRAFT_REPO_ROOT/jpezy_amd/bin
for t in 1 0; do
  if [ $t = 1 ]; then export JPEZY_IO_THREADS=1; else unset JPEZY_IO_THREADS; fi
  echo "== JPEZY_IO_THREADS=${JPEZY_IO_THREADS:-all}"
  for i in 1 2; do { time $B/jpezy_encode /tmp/big.ppm /tmp/big.jpg ; } 2>&1 | grep -i "time\|real\|user\|size"; done
done
for i in 1 2; do { time $B/jpezy_decode /tmp/big.jpg /tmp/back.ppm ; } 2>&1 | grep -i "time\|real\|user"; done
ls -la /tmp/big.ppm /tmp/big.jpg /tmp/back.ppm
