#!/usr/bin/env python3
"""Differential fuzz of the GPU Huffman decoder against the host decoder (and of the GPU entropy coder against the host writer): random sizes, contents, qualities, sampling
factors, optimised tables, restart intervals (libjpeg via PIL) and jpezy's own encoder; every scan goes to the GPU decoder (min_bytes 0)."""
import io
import os
import sys
import time
from pathlib import Path

import numpy as np
from PIL import Image, ImageFile

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests" / "fuzz"))
import jpezy_amd as J  # noqa: E402
from run_host_fuzz import mutate  # noqa: E402

ImageFile.MAXBLOCK = 1 << 24


def content(rng, H, W, kind):
    yy, xx = np.mgrid[0:H, 0:W]
    if kind == 0:
        return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    if kind == 1:   # smooth + noise
        base = (np.sin(xx / 37.0) * 60 + np.cos(yy / 23.0) * 50 + 128)
        return np.clip(base[..., None] + rng.normal(0, 6, (H, W, 3)), 0, 255).astype(np.uint8)
    if kind == 2:   # blobs and edges
        img = np.zeros((H, W, 3), np.float64)
        for _ in range(20):
            cx, cy, rad = rng.integers(0, W), rng.integers(0, H), rng.integers(5, max(6, min(H, W) // 3))
            img[(xx - cx) ** 2 + (yy - cy) ** 2 < rad * rad] = rng.integers(0, 256, 3)
        return np.clip(img + rng.normal(0, 2, img.shape), 0, 255).astype(np.uint8)
    if kind == 3:   # flat with a few dots: almost periodic stream
        img = np.full((H, W, 3), rng.integers(0, 256), np.uint8)
        img[rng.integers(0, H, 30), rng.integers(0, W, 30)] = 255
        return img
    return (rng.integers(0, 4, (H, W, 3)) * 64 + 20).astype(np.uint8)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    n_mut = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # damaged copies per case: GPU and host decoder must agree on them too
    mut_ok = mut_err = 0
    ctx = J.Context(0)
    ctx.set_huffdec_min_bytes(0)
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 2026)
    gpu = host = 0
    t_gpu, passes, fell = 0.0, {}, []
    for case in range(n_cases):
        W, H = int(rng.integers(8, 1400)), int(rng.integers(8, 1100))
        kind = int(rng.integers(0, 5))
        img = content(rng, H, W, kind)
        files = []
        kw = dict(quality=int(rng.integers(5, 100)), subsampling=int(rng.integers(0, 3)), optimize=bool(rng.integers(0, 2)))
        if rng.integers(0, 4) == 0:                                        # restart intervals: rows of MCUs or a few MCUs
            if rng.integers(0, 2):
                kw["restart_marker_rows"] = int(rng.integers(1, 4))
            else:
                kw["restart_marker_blocks"] = int(rng.integers(1, 40))
        buf = io.BytesIO(); Image.fromarray(img).save(buf, "JPEG", **kw); files.append(("pil", kw, buf.getvalue()))
        if rng.integers(0, 3) == 0:
            buf = io.BytesIO(); Image.fromarray(img[..., 0]).save(buf, "JPEG", quality=kw["quality"]); files.append(("pil-gray", kw, buf.getvalue()))
        r, g, b = (np.ascontiguousarray(img[..., k]).reshape(-1) for k in range(3))
        gray = bool(rng.integers(0, 2))
        own = ctx.encode_jpeg(r, g, b, W, H, gray=gray)               # FDCT + GPU entropy coder
        if own != J.write_jpeg(ctx.fdct_quant(r, g, b, W, H, gray=gray), W, H, gray):   # same coefficients through the host writer
            print("ENCODER MISMATCH", case, W, H, gray)
            return 1
        files.append(("jpezy", {}, own))
        for name, kw, data in files:
            info, want = J.read_jpeg(data)
            t0 = time.perf_counter()
            _, got = ctx.read_jpeg_gpu(data)
            t_gpu += time.perf_counter() - t0
            if ctx.last_huffdec_passes():
                gpu += 1
                passes[ctx.last_huffdec_passes()] = passes.get(ctx.last_huffdec_passes(), 0) + 1
            else:
                host += 1
                fell.append(f"case {case} {name} {W}x{H} kind {kind} {len(data)} B")
            if not np.array_equal(got.cpu().numpy(), want):
                print("MISMATCH", case, name, W, H, kw)
                return 1
        for m in range(n_mut):
            # damaged files: same verdict and, when they decode, the same coefficients (the GPU decoder hands anything
            # irregular to the host decoder; what it keeps must be what the host decoder would have produced)
            data = mutate(files[m % len(files)][2], rng)
            try:
                _, want = J.read_jpeg(data)
            except J.JpezyError:
                want = None
            try:
                _, got = ctx.read_jpeg_gpu(data)
                got = got.cpu().numpy()
            except J.JpezyError:
                got = None
            if (want is None) != (got is None) or (want is not None and not np.array_equal(got, want)):
                print("MUTANT MISMATCH", case, m, W, H, "host", "error" if want is None else "ok", "gpu", "error" if got is None else "ok")
                Path("gpurun_out").mkdir(exist_ok=True)
                Path(f"gpurun_out/mutant_{case}_{m}.jpg").write_bytes(data)
                return 1
            if want is None:
                mut_err += 1
            else:
                mut_ok += 1
    print(f"{gpu + host} files identical ({gpu} decoded by the GPU decoder, {host} handed to the host decoder)"
          + (f"; {mut_ok + mut_err} damaged copies: {mut_ok} decode identically, {mut_err} rejected by both" if n_mut else ""))
    print(f"jpezy_read_jpeg_gpu time over the intact files: {t_gpu * 1e3:.1f} ms; synchronisation launches per file (launches: files): "
          f"{dict(sorted(passes.items()))}; library {os.environ.get('JPEZY_LIB', 'in-tree')}")
    for line in fell:
        print("  host decoder:", line)
    return 0


if __name__ == "__main__":
    sys.exit(main())
