#!/usr/bin/env python3
"""jpezy_decode_jpeg (.jpg bytes on the host -> planes on the host) through bare ctypes with preallocated, touched output planes -- the
Python wrapper's fresh numpy arrays cost more in page faults than the call does -- so that builds of the library can be timed side by
side on one box:   python tools/measure/measure_decode_single_raw.py path/to/libjpezy_a.so [path/to/libjpezy_b.so ...]"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402
from jpezy_amd import api  # noqa: E402

mk = J.Context(0)
files = {}
for (W, H) in ((4096, 4096), (1920, 1080)):
    rng = np.random.default_rng(1)
    r, g, b = (rng.integers(0, 256, W * H, dtype=np.uint8) for _ in range(3))
    files[f"{W}x{H} random pixels"] = (W, H, np.frombuffer(mk.encode_jpeg(r, g, b, W, H), dtype=np.uint8).copy())
    yy, xx = np.mgrid[0:H, 0:W]
    sm = ((xx * 3 + yy * 2) // 8 % 256).astype(np.uint8).reshape(-1)
    files[f"{W}x{H} smooth"] = (W, H, np.frombuffer(mk.encode_jpeg(sm, sm[::-1].copy(), np.roll(sm, 77), W, H), dtype=np.uint8).copy())
ref = {name: mk.decode_jpeg(a.tobytes()) for name, (W, H, a) in files.items()}
mk.close()

vp = C.c_void_p
for path in sys.argv[1:]:
    lib = C.CDLL(path)
    lib.jpezy_ctx_create.restype = vp
    lib.jpezy_ctx_create.argtypes = [C.c_int]
    lib.jpezy_ctx_destroy.argtypes = [vp]
    lib.jpezy_decode_jpeg.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp, vp, vp, C.c_size_t]
    ctx = lib.jpezy_ctx_create(0)
    for name, (W, H, a) in files.items():
        info = api.FrameInfo()
        planes = [np.zeros(W * H, dtype=np.uint8) for _ in range(3)]

        def call():
            rc = lib.jpezy_decode_jpeg(ctx, a.ctypes.data, a.size, 0, C.byref(info), planes[0].ctypes.data, planes[1].ctypes.data,
                                       planes[2].ctypes.data, W * H)
            assert rc == 0, rc
        call()
        call()
        ts = []
        for _ in range(7):
            t = time.perf_counter()
            call()
            ts.append(time.perf_counter() - t)
        ok = all(np.array_equal(p, q) for p, q in zip(planes, ref[name][1:]))
        print(f"{Path(path).name}: jpezy_decode_jpeg {name} ({a.size / 1e6:.2f} MB), planes preallocated: median {np.median(ts) * 1e3:.2f} ms, "
              f"min {min(ts) * 1e3:.2f} ms = {W * H / np.median(ts) / 1e6:.0f} Mpx/s; identical: {ok}", flush=True)
    lib.jpezy_ctx_destroy(ctx)
