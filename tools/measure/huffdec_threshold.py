#!/usr/bin/env python3
"""Where the GPU Huffman decoder overtakes the host decoder (jpezy_ctx_set_huffdec_min_bytes): scans of growing size, random pixels
(longest scans per pixel) and smooth content, host decoder against the GPU decoder forced on."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402

ctx = J.Context(0)
ctx.set_huffdec_min_bytes(0)
for (W, H) in ((256, 256), (512, 384), (640, 480), (800, 600), (1024, 768), (1280, 960), (1920, 1080)):
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:H, 0:W]
    sm = np.clip((xx * 3 + yy * 2) // 8 % 256 + rng.normal(0, 10, (H, W)), 0, 255).astype(np.uint8).reshape(-1)
    for name, planes in (("random", [rng.integers(0, 256, W * H, dtype=np.uint8) for _ in range(3)]), ("smooth+noise", [sm, sm[::-1].copy(), np.roll(sm, 77)])):
        data = ctx.encode_jpeg(*planes, W, H)
        arr = np.frombuffer(data, dtype=np.uint8).copy()
        info, want = J.read_jpeg(data)
        co = torch.empty(want.size, dtype=torch.int16, device="cuda:0")
        th = min(_t for _t in (lambda: [(lambda t0: (J.read_jpeg(data), time.perf_counter() - t0)[1])(time.perf_counter()) for _ in range(5)])())
        ctx.read_jpeg_gpu_into(arr, co)
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            ctx.read_jpeg_gpu_into(arr, co)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ok = np.array_equal(co.cpu().numpy().reshape(want.shape), want)
        print(f"{W}x{H} {name}: scan {len(data) / 1024:.0f} KiB; host {th * 1e3:.2f} ms, GPU {np.median(ts) * 1e3:.2f} ms ({ctx.last_huffdec_passes()} launches); identical {ok}", flush=True)
