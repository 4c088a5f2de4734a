#!/usr/bin/env python3
"""Differential fuzz of the GPU entropy coder (one coding pass: in-place LDS rows, ownership-merged tile streams, assembly,
byte stuffing) against the host writer: random frame sizes, coefficient densities and magnitudes -- chains of blocks shorter
than a word, blocks that overflow their LDS row, tiles of a single block, gray mode's zero chroma blocks, streams full of
0xFF --, batches, through both the host-delivered and the device-resident entry point.   usage: fuzz_entropy.py [cases] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402


def coeffs(rng, nblk):
    kind = rng.integers(0, 7)
    co = np.zeros((nblk, 64), np.int16)
    if kind == 0:                                       # DC only, tiny differences: 6..12-bit blocks, chains inside one word
        co[:, 0] = rng.integers(-2, 3, nblk)
    elif kind == 1:                                     # sparse
        dens = rng.uniform(0.01, 0.2)
        mask = rng.random((nblk, 64)) < dens
        co[mask] = rng.integers(-15, 16, int(mask.sum()))
    elif kind == 2:                                     # dense, every magnitude: rows overflow
        co[:] = rng.integers(-1023, 1024, (nblk, 64))
    elif kind == 3:                                     # all ones bits: 0xFF bytes
        co[:, 0] = -1023
        co[:, 1:] = 1023
    elif kind == 4:                                     # mixture per block
        for b in range(nblk):
            d = rng.choice([0.0, 0.03, 0.3, 1.0])
            mag = int(rng.choice([1, 7, 127, 1023]))
            mask = rng.random(64) < d
            co[b, mask] = rng.integers(-mag, mag + 1, int(mask.sum()))
    elif kind == 5:                                     # long zero runs: ZRL chains, value at the very end
        co[:, 0] = rng.integers(-1023, 1024, nblk)
        pos = rng.integers(17, 64, nblk)
        co[np.arange(nblk), pos] = rng.integers(1, 1024, nblk)
    else:                                               # low-frequency heavy, like pictures
        scale = np.maximum(1, (300 / (1 + np.arange(64)) ** 1.5)).astype(np.int64)
        co[:] = (rng.normal(0, 1, (nblk, 64)) * scale).astype(np.int16).clip(-1023, 1023)
    return co


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctx = J.Context(0)
    t0 = time.time()
    n_files = n_bytes = 0
    for case in range(n_cases):
        if case % 9 == 8:
            W, H = int(rng.integers(600, 1400)), int(rng.integers(600, 1400))
        else:
            W, H = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        gray = bool(rng.integers(0, 2))
        n = int(rng.integers(1, 4))
        mc, mr = J.mcu_grid(W, H)
        bpm = 4 if gray else 6
        cos = np.stack([coeffs(rng, mc * mr * bpm).reshape(mr, mc, bpm, 64) for _ in range(n)])
        want = [J.write_jpeg(cos[f], W, H, gray) for f in range(n)]
        d = torch.from_numpy(cos).cuda()
        got = ctx.write_jpeg_gpu(d, W, H, gray=gray, n_frames=n)
        for f in range(n):
            assert got[f] == want[f], f"case {case}: host-delivered form differs ({W}x{H} gray={gray} frame {f})"
        stride = max(len(w) for w in want) + int(rng.integers(0, 64))
        out = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
        sizes = torch.zeros(n, dtype=torch.int64, device="cuda")
        ctx.write_jpeg_gpu_dev(d.reshape(n, -1), W, H, out, sizes, gray=gray, n_frames=n)
        torch.cuda.synchronize()
        for f in range(n):
            assert int(sizes[f]) == len(want[f]), f"case {case}: size {int(sizes[f])} != {len(want[f])}"
            assert out[f, :len(want[f])].cpu().numpy().tobytes() == want[f], f"case {case}: device-resident form differs"
        n_files += 2 * n
        n_bytes += 2 * sum(len(w) for w in want)
        if case % 50 == 49:
            print(f"{case + 1} cases, {n_files} files, {n_bytes / 1e6:.1f} MB, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz_entropy: {n_cases} cases, {n_files} files ({n_bytes / 1e6:.1f} MB) identical to the host writer, seed {seed}")
    ctx.close()


if __name__ == "__main__":
    main()
