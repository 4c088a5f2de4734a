#!/usr/bin/env python3
"""jpezy_decode_jpeg_batch of 256 x 1920x1080 files through bare ctypes, so that an older build of the library can be timed beside
the current one on the same box:   python tools/measure/measure_decode_batch_raw.py path/to/libjpezy_hip.so
The files are made with the in-tree library (random pixels: 0.66 MB scans; smooth + noise: smaller scans)."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402
from jpezy_amd import api  # noqa: E402

W, H, n = 1920, 1080, 256
mk = J.Context(0)
sets = {}
yy, xx = np.mgrid[0:H, 0:W]
for name in ("random pixels", "smooth + noise"):
    base = []
    for k in range(16):
        rng = np.random.default_rng(1000 + k)
        if name == "random pixels":
            planes = [rng.integers(0, 256, W * H, dtype=np.uint8) for _ in range(3)]
        else:
            p = np.clip((xx * (2 + k % 3) + yy * 3) // 8 % 256 + rng.normal(0, 6, (H, W)), 0, 255).astype(np.uint8).reshape(-1)
            planes = [p, p[::-1].copy(), np.roll(p, 77)]
        base.append(mk.encode_jpeg(*planes, W, H))
    sets[name] = [base[k % 16] for k in range(n)]
# other layouts (libjpeg): 4:4:4 and 4:2:2 files of picture-like content -- the batch form of the generic kernels (round 3)
import io  # noqa: E402
from PIL import Image, ImageFile  # noqa: E402
ImageFile.MAXBLOCK = 1 << 24
for name, sub in (("libjpeg 4:4:4 smooth + noise", 0), ("libjpeg 4:2:2 smooth + noise", 1)):
    base = []
    for k in range(16):
        rng = np.random.default_rng(2000 + k)
        img = np.clip((np.sin(xx / (30.0 + k)) * 60 + np.cos(yy / 23.0) * 50 + 128)[..., None] + rng.normal(0, 8, (H, W, 3)), 0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", quality=85, subsampling=sub)
        base.append(buf.getvalue())
    sets[name] = [base[k % 16] for k in range(n)]
ref = {name: mk.decode_jpeg(fl[5]) for name, fl in sets.items()}
mk.close()

lib = C.CDLL(sys.argv[1])
lib.jpezy_ctx_create.restype = C.c_void_p
vp = C.c_void_p
lib.jpezy_decode_jpeg_batch.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
ctx = lib.jpezy_ctx_create(0)
for name, fl in sets.items():
    arrs = [np.frombuffer(f, dtype=np.uint8) for f in fl]
    planes = [[np.zeros(W * H, dtype=np.uint8) for _ in range(3)] for _ in range(n)]
    vpa = C.c_void_p * n
    data = vpa(*[a.ctypes.data for a in arrs]); lens = (C.c_size_t * n)(*[a.size for a in arrs])
    rr, gg, bb = (vpa(*[p[k].ctypes.data for p in planes]) for k in range(3))
    caps = (C.c_size_t * n)(*[W * H] * n); status = (C.c_int * n)(); infos = (api.FrameInfo * n)()

    def call():
        assert lib.jpezy_decode_jpeg_batch(ctx, n, data, lens, 0, infos, rr, gg, bb, caps, status) == 0
    call()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); call(); ts.append(time.perf_counter() - t)
    tb = min(ts)
    fast = lib.jpezy_ctx_last_batch_fast_count(C.c_void_p(ctx)) if hasattr(lib, "jpezy_ctx_last_batch_fast_count") else "n/a"
    ok = all(np.array_equal(planes[5][k], ref[name][1 + k]) for k in range(3)) and np.array_equal(planes[5 + 16 * 15][1], ref[name][2])
    print(f"{sys.argv[1].split('/')[-1]}: 256 x 1920x1080 {name} ({len(fl[0]) / 1e6:.2f} MB each): {tb * 1e3:.1f} ms = {tb * 1e3 / n:.3f} ms/file "
          f"({W * H * n / tb / 1e6:.0f} Mpx/s); files through the batch form: {fast}; spot checks identical: {ok}")
