#!/usr/bin/env python3
"""Development probe: Huffman decoding of a 4096x4096 random-pixel .jpg on the GPU vs on the host."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402


def main():
    ctx = J.Context(0)
    for (W, H) in ((4096, 4096), (1920, 1080)):
        rng = np.random.default_rng(1)
        r, g, b = (rng.integers(0, 256, W * H, dtype=np.uint8) for _ in range(3))
        jpg = ctx.encode_jpeg(r, g, b, W, H)
        yy, xx = np.mgrid[0:H, 0:W]
        sm = ((xx * 3 + yy * 2) // 8 % 256).astype(np.uint8).reshape(-1)
        jpg_s = ctx.encode_jpeg(sm, sm[::-1].copy(), np.roll(sm, 77), W, H)
        for name, data in (("random pixels", jpg), ("smooth", jpg_s)):
            info, want = J.read_jpeg(data)
            t = time.perf_counter(); J.read_jpeg(data); th = time.perf_counter() - t
            ctx.read_jpeg_gpu(data); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                ginfo, got = ctx.read_jpeg_gpu(data)
            torch.cuda.synchronize()
            tg = (time.perf_counter() - t) / 5
            ok = np.array_equal(got.cpu().numpy(), want)
            t = time.perf_counter()
            for _ in range(3):
                ctx.decode_jpeg(data)
            te = (time.perf_counter() - t) / 3
            print(f"{W}x{H} {name}: jpezy_decode_jpeg (.jpg bytes on the host -> r,g,b planes on the host, Huffman + IDCT + colour on the GPU): {te * 1e3:.2f} ms")
            print(f"{W}x{H} {name}: {len(data) / 1e6:.2f} MB; GPU Huffman decode {tg * 1e3:.2f} ms ({ctx.last_huffdec_passes()} passes) = "
                  f"{W * H / tg / 1e6:.0f} Mpx/s; host {th * 1e3:.1f} ms = {W * H / th / 1e6:.0f} Mpx/s; identical: {ok}")
    # files in other layouts (libjpeg): device Huffman decoder + the generic kernels
    import io
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = 1 << 26
    W = H = 4096
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.clip((np.sin(xx / 37.0) * 60 + np.cos(yy / 23.0) * 50 + 128)[..., None] + rng.normal(0, 12, (H, W, 3)), 0, 255).astype(np.uint8)
    for name, kw, im in (("4:4:4", dict(subsampling=0, quality=90), img), ("4:2:2", dict(subsampling=1, quality=90), img),
                         ("one component", dict(quality=90), img[..., 0])):
        buf = io.BytesIO()
        Image.fromarray(im).save(buf, "JPEG", **kw)
        data = buf.getvalue()
        ctx.decode_jpeg(data)
        t = time.perf_counter()
        for _ in range(3):
            ctx.decode_jpeg(data)
        te = (time.perf_counter() - t) / 3
        print(f"{W}x{H} libjpeg {name} ({len(data) / 1e6:.1f} MB): jpezy_decode_jpeg {te * 1e3:.2f} ms ({ctx.last_huffdec_passes()} passes)")

    # restart intervals (DRI): every interval an independent stream on the device
    for name, kw, im in (("4:2:0, an interval per MCU row", dict(subsampling=2, quality=90, restart_marker_rows=1), img),
                         ("4:4:4, an interval per 8 MCUs", dict(subsampling=0, quality=90, restart_marker_blocks=8), img)):
        buf = io.BytesIO()
        Image.fromarray(im).save(buf, "JPEG", **kw)
        data = buf.getvalue()
        info, want = J.read_jpeg(data)
        t = time.perf_counter(); J.read_jpeg(data); th = time.perf_counter() - t
        ctx.read_jpeg_gpu(data); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            ginfo, got = ctx.read_jpeg_gpu(data)
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t) / 5
        ok = np.array_equal(got.cpu().numpy(), want)
        print(f"{W}x{H} libjpeg {name} ({len(data) / 1e6:.1f} MB, restart interval {info.restart_interval}): GPU Huffman decode {tg * 1e3:.2f} ms "
              f"({'GPU' if ctx.last_huffdec_passes() else 'HOST'}); host {th * 1e3:.1f} ms; identical: {ok}")

    # a batch of different 1080p files: one call per file vs jpezy_decode_jpeg_batch (up to 8 files in flight)
    W, H = 1920, 1080
    files = []
    for k in range(32):
        rng = np.random.default_rng(100 + k)
        r, g, b = (rng.integers(0, 256, W * H, dtype=np.uint8) for _ in range(3))
        files.append(ctx.encode_jpeg(r, g, b, W, H))
    ctx.set_huffdec_min_bytes(0)
    ctx.decode_jpeg_batch(files[:8])
    t = time.perf_counter()
    single = [ctx.decode_jpeg(f) for f in files]
    ts = time.perf_counter() - t
    t = time.perf_counter()
    batch = ctx.decode_jpeg_batch(files)
    tb = time.perf_counter() - t
    same = all(np.array_equal(a[1], b_[1]) and np.array_equal(a[3], b_[3]) for a, b_ in zip(single, batch))
    print(f"32 x 1920x1080 random pixels ({len(files[0]) / 1e6:.2f} MB each), .jpg on the host -> planes on the host: one call per file "
          f"{ts * 1e3 / 32:.2f} ms/file, jpezy_decode_jpeg_batch {tb * 1e3 / 32:.2f} ms/file ({W * H * 32 / tb / 1e6:.0f} Mpx/s, "
          f"{ctx.last_batch_fast_count()} files through the batch form of the kernels); identical: {same}")
    # 256 x 1080p through the C-ABI with preallocated, touched output planes (the Python wrapper's allocations
    # are not part of the path); random pixels (0.66 MB scans) and picture-like content (smaller scans)
    import ctypes as C
    from jpezy_amd import api
    lib = api.load_library()
    for name, make in (("random pixels", lambda k: [np.random.default_rng(1000 + k).integers(0, 256, W * H, dtype=np.uint8) for _ in range(3)]),
                       ("smooth + noise", lambda k: [np.clip(((np.mgrid[0:H, 0:W][1] * (2 + k % 3) + np.mgrid[0:H, 0:W][0] * 3) // 8 % 256
                                                              + np.random.default_rng(k).normal(0, 6, (H, W))), 0, 255).astype(np.uint8).reshape(-1)] * 3)):
        base = [ctx.encode_jpeg(*make(k), W, H) for k in range(16)]
        n = 256
        fl = [base[k % 16] for k in range(n)]
        arrs = [np.frombuffer(f, dtype=np.uint8) for f in fl]
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
        tb = min(ts)
        ref = ctx.decode_jpeg(fl[5])
        ok = np.array_equal(planes[5][0], ref[1]) and np.array_equal(planes[255][2], ctx.decode_jpeg(fl[255])[3])
        print(f"256 x 1920x1080 {name} ({len(fl[0]) / 1e6:.2f} MB each): jpezy_decode_jpeg_batch {tb * 1e3:.1f} ms = {tb * 1e3 / n:.3f} ms/file "
              f"({W * H * n / tb / 1e6:.0f} Mpx/s; {ctx.last_batch_fast_count()} files through the batch form); spot checks identical: {ok}")


if __name__ == "__main__":
    main()
