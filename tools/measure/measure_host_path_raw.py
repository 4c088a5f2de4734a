#!/usr/bin/env python3
"""The host-buffer FDCT entry point of ANY build of the library through bare ctypes (only symbols every round exported), so
that an older libjpezy_hip.so can be timed beside the current one on the same box:
    python tools/measure/measure_host_path_raw.py path/to/libjpezy_hip.so"""
import ctypes as C
import sys
import time

import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: bind to torch's)

lib = C.CDLL(sys.argv[1])
lib.jpezy_ctx_create.restype = C.c_void_p
lib.jpezy_fdct_quant.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
ctx = lib.jpezy_ctx_create(0)
p8 = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
rng = np.random.default_rng(0)
for (W, H, F) in ((4096, 4096, 1), (1920, 1080, 32)):
    r, g, b = (rng.integers(0, 256, W * H * F, dtype=np.uint8) for _ in range(3))
    ncoef = ((W + 15) // 16) * ((H + 15) // 16) * 6 * 64 * F
    out = np.empty(ncoef, dtype=np.int16)
    for _ in range(2):
        assert lib.jpezy_fdct_quant(ctx, p8(r), p8(g), p8(b), W, H, 0, F, p8(out)) == 0
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        assert lib.jpezy_fdct_quant(ctx, p8(r), p8(g), p8(b), W, H, 0, F, p8(out)) == 0
        ts.append(time.perf_counter() - t)
    tf = []
    for _ in range(5):
        rr, gg, bb, oo = r.copy(), g.copy(), b.copy(), np.empty(ncoef, dtype=np.int16)
        t = time.perf_counter()
        assert lib.jpezy_fdct_quant(ctx, p8(rr), p8(gg), p8(bb), W, H, 0, F, p8(oo)) == 0
        tf.append(time.perf_counter() - t)
    print(f"{sys.argv[1].split('/')[-1]}: jpezy_fdct_quant {F} x {W}x{H}: buffers reused {min(ts)*1e3:.2f} ms, fresh buffers {min(tf)*1e3:.2f} ms "
          f"({W*H*F/min(tf)/1e6:.0f} Mpx/s)")
