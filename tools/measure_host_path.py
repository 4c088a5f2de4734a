#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points and the host serial tail (for DESIGN.md; never `value`)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import jpezy_amd as J  # noqa: E402


def best(fn, n=5):
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return min(ts)


def main():
    W = H = 4096
    rng = np.random.default_rng(0)
    r, g, b = (rng.integers(0, 256, W * H, dtype=np.uint8) for _ in range(3))
    ctx = J.Context(0)
    co = ctx.fdct_quant(r, g, b, W, H)
    t = best(lambda: ctx.fdct_quant(r, g, b, W, H))
    print(f"jpezy_fdct_quant host buffers 4096x4096 (H2D 50 MB + kernel + D2H 50 MB, pageable): {t*1e3:.2f} ms = {W*H/t/1e6:.0f} Mpx/s")
    t = best(lambda: ctx.dequant_idct(co, W, H))
    print(f"jpezy_dequant_idct host buffers 4096x4096: {t*1e3:.2f} ms = {W*H/t/1e6:.0f} Mpx/s")
    t = best(lambda: J.write_jpeg(co, W, H), 3)
    jpg = J.write_jpeg(co, W, H)
    print(f"jpezy_write_jpeg (host Huffman+JFIF, 1 thread) 4096x4096 random pixels: {t*1e3:.1f} ms = {W*H/t/1e6:.0f} Mpx/s, {len(jpg)/1e6:.1f} MB")
    t = best(lambda: J.read_jpeg(jpg), 3)
    print(f"jpezy_read_jpeg (host parse+Huffman decode, 1 thread): {t*1e3:.1f} ms = {W*H/t/1e6:.0f} Mpx/s")


if __name__ == "__main__":
    main()
