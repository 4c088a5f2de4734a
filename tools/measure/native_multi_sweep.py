#!/usr/bin/env python3
"""Sweep of the native multi-GPU entry's two tuning knobs on one device: feeder threads per lane and frames per chunk.
256 frames 1920x1080, pageable host planes -> .jpg in host memory, handle outside the bracket, median of 3 calls.
    python tools/measure/native_multi_sweep.py  > gpurun_out/native_multi_sweep.txt"""
import ctypes as C
import statistics
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402
from jpezy_amd import api  # noqa: E402

lib = api.load_library()
W, H, F = 1920, 1080, 256
plane = W * H
rng = np.random.default_rng(1)
base = [rng.integers(0, 256, (4, plane), dtype=np.uint8) for _ in range(3)]
planes = [np.ascontiguousarray(np.tile(b, (F // 4, 1))).reshape(-1) for b in base]
stride = 1 << 20
jpg = np.zeros(F * stride, np.uint8)
sizes = (C.c_longlong * F)()
out = api.MultiOut()
out.jpg, out.jpg_stride, out.jpg_sizes = jpg.ctypes.data, stride, sizes
devs = (C.c_int * 1)(0)
print(f"{'chunk':>5} {'feeders':>7} {'ms':>8} {'Gpx/s':>7} {'GB/s up':>8}")
for chunk in (1, 2, 4, 8):
    h = lib.jpezy_multi_create(devs, 1, W, H, 0, chunk)
    assert h, lib.jpezy_hip_last_error()
    for nf in (2, 3, 4, 5, 6):
        assert lib.jpezy_multi_set_feeder_threads(h, nf) == 0
        ts = []
        for i in range(4):
            t0 = time.perf_counter()
            rc = lib.jpezy_multi_encode(h, *[api._np_ptr(p) for p in planes], F, b"x", C.byref(out))
            dt = time.perf_counter() - t0
            assert rc == 0, lib.jpezy_hip_last_error()
            if i:
                ts.append(dt)
        dt = statistics.median(ts)
        print(f"{chunk:5d} {nf:7d} {dt * 1e3:8.2f} {F * plane / dt / 1e9:7.2f} {3 * plane * F / dt / 1e9:8.2f}", flush=True)
    lib.jpezy_multi_destroy(h)
