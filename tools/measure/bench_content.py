#!/usr/bin/env python3
"""Kernel time as a function of image CONTENT (the exact paths are data dependent): random pixels (the benchmark
workload), smooth gradients + mild noise (photo-like: sparse high frequencies), flat colour, a 2-level checkerboard
(worst case for the rational-coefficient guard).  4096x4096, inputs resident in HBM, HIP events."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402


def images(W, H):
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    yield "random", [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(3)]
    base = 128 + 90 * np.sin(xx / 97.0) * np.cos(yy / 61.0)
    yield "smooth+noise", [np.clip(base * s + rng.normal(0, 3, (H, W)), 0, 255).astype(np.uint8) for s in (1.0, 0.9, 0.8)]
    yield "smooth", [np.clip(base * s, 0, 255).astype(np.uint8) for s in (1.0, 0.9, 0.8)]
    yield "flat", [np.full((H, W), v, np.uint8) for v in (200, 100, 50)]
    cb = (((xx.astype(np.int32) // 4) + (yy.astype(np.int32) // 4)) % 2 * 215 + 20).astype(np.uint8)
    yield "checker4", [cb, cb, cb]
    two = (rng.integers(0, 2, (H, W)) * 255).astype(np.uint8)
    yield "2-level noise", [two, np.roll(two, 1), two[::-1].copy()]
    four = (rng.integers(0, 4, (H, W)) * 64 + 31).astype(np.uint8)
    yield "4-level noise", [four, four, four]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    W = H = 4096
    dev = torch.device("cuda", 0)
    ctx = J.Context(0)
    for name, planes in images(W, H):
        d = [torch.from_numpy(p.reshape(-1)).to(dev) for p in planes]
        co = torch.empty(J.coeff_count(W, H), dtype=torch.int16, device=dev)
        out = [torch.empty(W * H, dtype=torch.uint8, device=dev) for _ in range(3)]
        res = []
        for variant in (1, 0):
            ctx.set_variant(variant)
            ctx.fallback_count()
            t = timeit(lambda: ctx.fdct_quant_dev(d[0], d[1], d[2], W, H, co))
            res.append((t, ctx.fallback_count() / 23))
        ctx.set_variant(1)
        ctx.fdct_quant_dev(d[0], d[1], d[2], W, H, co)
        torch.cuda.synchronize()
        ctx.fallback_count()
        td = timeit(lambda: ctx.dequant_idct_dev(co, W, H, out[0], out[1], out[2]))
        nd = ctx.fallback_count() / 23
        nz = float((co != 0).float().mean())
        print(f"{name:14s} encode v1 {res[0][0]:8.1f} us ({res[0][1]:9.0f} resolves)  v0 {res[1][0]:8.1f} us ({res[1][1]:9.0f})  "
              f"decode {td:8.1f} us ({nd:10.0f} exact samples)  nonzero coeffs {nz*100:5.1f} %", flush=True)


if __name__ == "__main__":
    main()
