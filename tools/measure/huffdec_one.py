#!/usr/bin/env python3
"""Development probe: 30 calls of jpezy_read_jpeg_gpu on one W x H random-pixel file (for a rocprofv3 --kernel-trace timeline of the
single-file path's launches and gaps).  python tools/measure/huffdec_one.py [W] [H]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = int(sys.argv[2]) if len(sys.argv) > 2 else 768
ctx = J.Context(0)
ctx.set_huffdec_min_bytes(0)
rng = np.random.default_rng(3)
data = ctx.encode_jpeg(*[rng.integers(0, 256, W * H, dtype=np.uint8) for _ in range(3)], W, H)
arr = np.frombuffer(data, dtype=np.uint8).copy()
co = torch.empty(J.coeff_count(W, H), dtype=torch.int16, device="cuda:0")
for _ in range(5):
    ctx.read_jpeg_gpu_into(arr, co)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(30):
    ctx.read_jpeg_gpu_into(arr, co)
torch.cuda.synchronize()
print(f"{W}x{H}: {len(data) / 1024:.0f} KiB, {(time.perf_counter() - t) / 30 * 1e3:.3f} ms per call")
