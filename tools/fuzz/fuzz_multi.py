#!/usr/bin/env python3
"""Differential fuzz of the native multi-GPU entry (jpezy_multi_create / jpezy_multi_encode; jpezy_encode_batch_multi) against the
single-frame entry points on ONE device: random frame sizes (ragged, tiny, a few large), frame counts, lane counts (several lanes
on the one device index), chunk sizes and feeder counts, gray / colour, host or root-device delivery, coefficients and / or files,
pageable or caller-pinned planes, file strides that are too small for some frames, handles reused across calls of changing length.
Every frame of every call must equal jpezy_fdct_quant / jpezy_encode_jpeg of the same planes.   usage: fuzz_multi.py [cases] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctx = J.Context(0)
    t0 = time.time()
    frames_done = calls = 0
    for case in range(n_cases):
        if case % 11 == 10:
            W, H = int(rng.integers(500, 2100)), int(rng.integers(300, 1200))
        else:
            W, H = int(rng.integers(1, 260)), int(rng.integers(1, 200))
        gray = bool(rng.integers(0, 2))
        lanes = int(rng.integers(1, 6))
        chunk = int(rng.choice([0, 1, 2, 3, 5]))
        fmax = 6 if W * H > 300000 else 40
        pool = int(rng.integers(1, fmax + 1))
        kind = rng.integers(0, 3)
        if kind == 0:
            px = [rng.integers(0, 256, (pool, W * H), dtype=np.uint8) for _ in range(3)]
        elif kind == 1:                                                        # flat and near-flat frames: the shortest files
            px = [np.repeat(rng.integers(0, 256, (pool, 1), dtype=np.uint8), W * H, axis=1) for _ in range(3)]
        else:                                                                  # smooth + noise
            g = (np.linspace(0, 255, W * H)[None, :] + rng.normal(0, 6, (pool, W * H))).clip(0, 255).astype(np.uint8)
            px = [g, np.roll(g, 7, axis=1), 255 - g]
        want_co = [ctx.fdct_quant(px[0][f], px[1][f], px[2][f], W, H, gray=gray).reshape(-1) for f in range(pool)]
        want_jpg = [ctx.encode_jpeg(px[0][f], px[1][f], px[2][f], W, H, gray=gray) for f in range(pool)]
        longest = max(len(j) for j in want_jpg)
        one_shot = case % 5 == 4
        M = None if one_shot else J.MultiEncoder([0] * lanes, W, H, gray=gray, chunk_frames=chunk)
        if M is not None and rng.integers(0, 2):
            M.set_feeder_threads(int(rng.integers(1, 7)))
        try:
            for _ in range(1 if one_shot else int(rng.integers(1, 5))):
                n = int(rng.integers(1, pool + 1))
                sel = rng.integers(0, pool, n)
                planes = [np.ascontiguousarray(p[sel]).reshape(-1) for p in px]
                if rng.integers(0, 4) == 0:
                    planes = [torch.from_numpy(p).pin_memory() for p in planes]
                on_root = bool(rng.integers(0, 2))
                co_too = bool(rng.integers(0, 2))
                jpg_too = bool(rng.integers(0, 4)) or not co_too
                tight = jpg_too and rng.integers(0, 5) == 0                    # a stride some files do not fit
                stride = max(700, longest - int(rng.integers(1, max(2, longest // 3)))) if tight else None
                if one_shot:
                    co, jpg = J.encode_batch_multi([0] * lanes, *planes, W, H, n, gray=gray, chunk_frames=chunk, want_coeffs=co_too,
                                                   want_jpg=jpg_too, on_root_device=on_root, jpg_stride=stride)
                else:
                    co, jpg = M.encode(*planes, n, want_coeffs=co_too, want_jpg=jpg_too, on_root_device=on_root, jpg_stride=stride)
                for i, f in enumerate(sel):
                    if co_too:
                        assert np.array_equal(co[i], want_co[f]), (case, "coeffs", W, H, gray, lanes, chunk, n, i)
                    if jpg_too:
                        if stride is not None and len(want_jpg[f]) > stride:
                            assert jpg[i] == -6, (case, "nospace", W, H, i, jpg[i] if isinstance(jpg[i], int) else len(jpg[i]))
                        else:
                            assert jpg[i] == want_jpg[f], (case, "jpg", W, H, gray, lanes, chunk, n, i, on_root)
                frames_done += n
                calls += 1
        finally:
            if M is not None:
                M.close()
        if case % 20 == 19:
            print(f"{case + 1} cases, {calls} calls, {frames_done} frames, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz_multi: {n_cases} cases, {calls} calls, {frames_done} frames identical to the single-frame entry points, seed {seed}")


if __name__ == "__main__":
    main()
