#!/usr/bin/env python3
"""One leg of tools/measure/scale_node.sh: the native multi-GPU entry (jpezy_multi_create / jpezy_multi_encode, ONE host process,
a lane per device, gather by hipMemcpyPeerAsync) on devices 0..N-1, a fresh process per leg, one JSON line on stdout.

    python tools/measure/native_multi_leg.py --gpus N [--on-root-device 0|1] [--frames-per-gpu 256] [--frames F] [--coeffs]

--frames: STRONG scaling (the same F frames whatever N; BASELINE configs[3] is 4096 -- 25 GB of host planes, so the default here is
--frames-per-gpu, weak scaling, and --frames 4096 is for the node that has the memory).  Host planes are pageable numpy memory (staged
through the lanes' pinned rings); the handle is created outside the bracket.  Never run on more than one GPU by the builder."""
import argparse
import ctypes as C
import json
import statistics
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402
from jpezy_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--on-root-device", type=int, default=0)
ap.add_argument("--frames-per-gpu", type=int, default=256)
ap.add_argument("--frames", type=int, default=0)
ap.add_argument("--coeffs", action="store_true", help="gather the coefficient buffers too (north_star's wording), not only the .jpg files")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()

import torch  # noqa: E402
W, H = 1920, 1080
plane = W * H
F = a.frames or a.frames_per_gpu * a.gpus
devs = list(range(a.gpus))
lib = api.load_library()
rng = np.random.default_rng(0x6A70)
base = [rng.integers(0, 256, (4, plane), dtype=np.uint8) for _ in range(3)]
planes = [np.ascontiguousarray(np.tile(b, (F // 4 + 1, 1))[:F]).reshape(-1) for b in base]
ctx = J.Context(0)
ref = ctx.encode_jpeg(base[0][1], base[1][1], base[2][1], W, H)
stride = 1 << 20
cpf = J.coeff_count(W, H, False)
sizes = (C.c_longlong * F)()
out = api.MultiOut()
out.jpg_stride, out.jpg_sizes, out.on_root_device = stride, sizes, a.on_root_device
keep = []
if a.on_root_device:
    dev = torch.device("cuda", 0)
    jpg = torch.zeros(F * stride, dtype=torch.uint8, device=dev)
    out.jpg = jpg.data_ptr()
    if a.coeffs:
        co = torch.empty(F * cpf, dtype=torch.int16, device=dev)
        out.coeffs = co.data_ptr()
else:
    jpg = np.zeros(F * stride, dtype=np.uint8)
    out.jpg = jpg.ctypes.data
    if a.coeffs:
        co = np.zeros(F * cpf, dtype=np.int16)
        out.coeffs = co.ctypes.data
darr = (C.c_int * len(devs))(*devs)
h = lib.jpezy_multi_create(darr, len(devs), W, H, 0, 0)
if not h:
    raise SystemExit(lib.jpezy_hip_last_error().decode(errors="replace"))
times = []
for i in range(a.reps + 1):
    t0 = time.perf_counter()
    rc = lib.jpezy_multi_encode(h, *[api._np_ptr(p) for p in planes], F, b"Encoded by jpezy", C.byref(out))
    if a.on_root_device:
        torch.cuda.synchronize(0)
    dt = time.perf_counter() - t0
    if rc != 0:
        raise SystemExit(lib.jpezy_hip_last_error().decode(errors="replace"))
    if i:
        times.append(dt)
st = (api.MultiLaneStats * len(devs))()
lib.jpezy_multi_last_stats(h, st, len(devs))
chunk = lib.jpezy_multi_chunk_frames(h)
lib.jpezy_multi_destroy(h)
dt = statistics.median(times)
ok = True
for f in (1, F - 3):
    got = (jpg[f * stride: f * stride + sizes[f]].cpu().numpy() if a.on_root_device else jpg[f * stride: f * stride + sizes[f]]).tobytes()
    ok = ok and got == ref
kernel_ms = max(s.kernel_ms for s in st)
print(json.dumps({
    "leg": "native_multi", "entry": "jpezy_multi_encode", "n_gpus": a.gpus, "devices": devs, "on_root_device": a.on_root_device,
    "gathers": "coefficients + .jpg" if a.coeffs else ".jpg", "frames": F, "chunk_frames": chunk, "scaling": "strong" if a.frames else "weak",
    "end_to_end_ms": round(dt * 1e3, 2), "end_to_end_ms_min_max": [round(min(times) * 1e3, 2), round(max(times) * 1e3, 2)],
    "Mpixels_per_s": round(F * plane / dt / 1e6, 1), "GBs_h2d_all_devices": round(3 * plane * F / dt / 1e9, 2),
    "kernel_only_ms_slowest_lane": round(kernel_ms, 2), "kernel_only_Mpixels_per_s": round(F * plane / (kernel_ms * 1e-3) / 1e6, 1) if kernel_ms else None,
    "gather_and_transfer_ms_not_hidden": round(max(0.0, dt * 1e3 - kernel_ms), 2),
    "lanes": [{"device": s.device, "frames": s.frames, "wall_ms": round(s.wall_ms, 2), "kernel_ms": round(s.kernel_ms, 2),
               "bytes_up": s.bytes_up, "bytes_down": s.bytes_down} for s in st],
    "equal_to_single_frame_entry": bool(ok)}))
