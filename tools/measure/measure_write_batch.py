#!/usr/bin/env python3
"""jpezy_write_jpeg_gpu_batch (device coefficients -> .jpg files in host memory) for a batch of 1080p frames: where the time goes."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402

W, H, F = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
ctx = J.Context(0)
lib = J.load_library()
gen = torch.Generator(device=dev); gen.manual_seed(1)
pr, pg, pb = (torch.randint(0, 256, (F, W * H), dtype=torch.uint8, device=dev, generator=gen) for _ in range(3))
co = torch.empty((F, J.coeff_count(W, H)), dtype=torch.int16, device=dev)
ctx.fdct_quant_dev(pr, pg, pb, W, H, co, n_frames=F)
torch.cuda.synchronize()
cap = lib.jpezy_jpeg_bound(W, H)
out = np.zeros(cap * F, dtype=np.uint8)
sizes = (C.c_long * F)()


def call():
    rc = lib.jpezy_write_jpeg_gpu_batch(ctx._h, co.data_ptr(), W, H, 0, F, b"Encoded by jpezy", out.ctypes.data, cap, sizes)
    assert rc == 0, rc


call()
ts = []
for _ in range(5):
    t = time.perf_counter(); call(); ts.append(time.perf_counter() - t)
total = sum(sizes[f] for f in range(F))
print(f"jpezy_write_jpeg_gpu_batch {F} x {W}x{H}: {min(ts) * 1e3:.2f} ms ({min(ts) * 1e3 / F:.3f} ms per frame), {total / 1e6:.1f} MB of .jpg, {W * H * F / min(ts) / 1e6:.0f} Mpx/s")
