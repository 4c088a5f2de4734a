#!/usr/bin/env python3
"""jpezy_decode_jpeg_batch on many SMALL files (thumbnails): the chain of launches per slice is latency, so the slice should hold more
files the smaller they are.  python tools/measure/measure_small_batch.py [n_files] [W] [H]   (JPEZY_BATCH_SLICE overrides the slice size)"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402
from jpezy_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ctx = J.Context(0)
ctx.set_huffdec_min_bytes(0)
yy, xx = np.mgrid[0:H, 0:W]
base = []
for k in range(32):
    rng = np.random.default_rng(k)
    p = np.clip((np.sin(xx / (9.0 + k)) * 60 + np.cos(yy / 13.0) * 50 + 128) + rng.normal(0, 10, (H, W)), 0, 255).astype(np.uint8).reshape(-1)
    base.append(ctx.encode_jpeg(p, p[::-1].copy(), np.roll(p, 31), W, H))
files = [base[k % 32] for k in range(n)]
lib = api.load_library()
arrs = [np.frombuffer(f, dtype=np.uint8) for f in files]
planes = [[np.zeros(W * H, dtype=np.uint8) for _ in range(3)] for _ in range(n)]
vpa = C.c_void_p * n
data = vpa(*[a.ctypes.data for a in arrs]); lens = (C.c_size_t * n)(*[a.size for a in arrs])
rr, gg, bb = (vpa(*[p[k].ctypes.data for p in planes]) for k in range(3))
caps = (C.c_size_t * n)(*[W * H] * n); status = (C.c_int * n)(); infos = (api.FrameInfo * n)()


def call():
    assert lib.jpezy_decode_jpeg_batch(ctx._h, n, data, lens, 0, infos, rr, gg, bb, caps, status) == 0


call()
ts = []
for _ in range(3):
    t = time.perf_counter(); call(); ts.append(time.perf_counter() - t)
ref = ctx.decode_jpeg(files[7])
ok = all(np.array_equal(planes[7][k], ref[1 + k]) for k in range(3))
print(f"{n} x {W}x{H} ({len(files[0]) / 1024:.0f} KiB each): {min(ts) * 1e3:.1f} ms = {min(ts) * 1e6 / n:.1f} us/file, {W * H * n / min(ts) / 1e6:.0f} Mpx/s; "
      f"through the batch form: {ctx.last_batch_fast_count()}; spot check identical: {ok}")
