#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points and the host serial tail (for DESIGN.md; never `value`)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
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
    import ctypes as C
    from jpezy_amd import api
    lib = api.load_library()
    out = np.empty(co.size, dtype=np.int16)
    p8 = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731

    def enc_same():       # the C-ABI call itself, caller's buffers reused (the Python wrapper's allocation is not part of the path)
        assert lib.jpezy_fdct_quant(ctx._h, p8(r), p8(g), p8(b), W, H, 0, 1, p8(out)) == 0
    t = best(enc_same)
    print(f"jpezy_fdct_quant host buffers 4096x4096 (H2D 50 MB + kernel + D2H 50 MB), caller reuses its buffers: {t*1e3:.2f} ms = {W*H/t/1e6:.0f} Mpx/s")

    def enc_fresh():      # fresh pages every call, as encoder::encode sees them (the object copies its planes at construction)
        rr, gg, bb = r.copy(), g.copy(), b.copy()
        oo = np.empty(co.size, dtype=np.int16)
        t0 = time.perf_counter()
        assert lib.jpezy_fdct_quant(ctx._h, p8(rr), p8(gg), p8(bb), W, H, 0, 1, p8(oo)) == 0
        return time.perf_counter() - t0
    t = min(enc_fresh() for _ in range(5))
    print(f"jpezy_fdct_quant host buffers 4096x4096, fresh buffers every call: {t*1e3:.2f} ms = {W*H/t/1e6:.0f} Mpx/s")
    planes = [np.empty(W * H, dtype=np.uint8) for _ in range(3)]
    qt, tq = api.annex_k_tables().qt, (C.c_uint8 * 3)(0, 1, 1)

    def dec_same():
        assert lib.jpezy_dequant_idct(ctx._h, p8(co), C.byref(qt), C.byref(tq), W, H, 0, 1, p8(planes[0]), p8(planes[1]), p8(planes[2])) == 0
    t = best(dec_same)
    print(f"jpezy_dequant_idct host buffers 4096x4096: {t*1e3:.2f} ms = {W*H/t/1e6:.0f} Mpx/s")
    # BASELINE configs[3] shard through the host path: 32 frames 1920x1080
    W2, H2, F = 1920, 1080, 32
    rb, gb, bb_ = (rng.integers(0, 256, W2 * H2 * F, dtype=np.uint8) for _ in range(3))
    ob = np.empty(J.coeff_count(W2, H2, False) * F, dtype=np.int16)

    def enc_batch():
        assert lib.jpezy_fdct_quant(ctx._h, p8(rb), p8(gb), p8(bb_), W2, H2, 0, F, p8(ob)) == 0
    t = best(enc_batch, 3)
    print(f"jpezy_fdct_quant host buffers 32 x 1920x1080: {t*1e3:.2f} ms = {W2*H2*F/t/1e6:.0f} Mpx/s")
    jb = np.empty(8 << 20, dtype=np.uint8)

    def enc_jpg():
        assert lib.jpezy_encode_jpeg(ctx._h, p8(r), p8(g), p8(b), W, H, 0, b"Encoded by jpezy", p8(jb), jb.size) > 0
    t = best(enc_jpg)
    print(f"jpezy_encode_jpeg host planes -> host .jpg 4096x4096: {t*1e3:.2f} ms = {W*H/t/1e6:.0f} Mpx/s")
    t = best(lambda: J.write_jpeg(co, W, H), 3)
    jpg = J.write_jpeg(co, W, H)
    print(f"jpezy_write_jpeg (host Huffman+JFIF, 1 thread) 4096x4096 random pixels: {t*1e3:.1f} ms = {W*H/t/1e6:.0f} Mpx/s, {len(jpg)/1e6:.1f} MB")
    t = best(lambda: J.read_jpeg(jpg), 3)
    print(f"jpezy_read_jpeg (host parse+Huffman decode, 1 thread): {t*1e3:.1f} ms = {W*H/t/1e6:.0f} Mpx/s")


if __name__ == "__main__":
    main()
