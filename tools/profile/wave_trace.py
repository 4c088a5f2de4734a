#!/usr/bin/env python3
"""Development probe: per-wave timeline of the f32 encode kernel (build with tools/ab/ab_build.py trace:-DJPEZY_TRACE,
run with JPEZY_LIB=ab/libjpezy_trace.so).  Each wave records s_memrealtime (100 MHz) at start, after its pixel loads
arrived, before its stores and at its end, plus HW_ID / XCC_ID."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402
from jpezy_amd import api  # noqa: E402


def main():
    W = H = 4096
    ctx = J.Context(0)
    lib = api.load_library()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(1)
    ring = 6
    planes = [torch.randint(0, 256, (ring, W * H), dtype=torch.uint8, device=dev, generator=g) for _ in range(3)]
    out = torch.empty((ring, J.coeff_count(W, H, False)), dtype=torch.int16, device=dev)
    for it in range(12):
        k = it % ring
        ctx.fdct_quant_dev(planes[0][k], planes[1][k], planes[2][k], W, H, out[k])
    torch.cuda.synchronize()
    n = 16384
    buf = np.zeros(n * 4, dtype=np.uint64)
    lib.jpezy_debug_read_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.jpezy_debug_read_trace(ctx._h, buf.ctypes.data, n * 4)
    t = buf.reshape(n, 4)
    t0 = t[:, 0].astype(np.int64); t0 -= t0.min()
    d_load = (t[:, 1] >> np.uint64(32)).astype(np.int64); d_comp = (t[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    d_end = t[:, 2].astype(np.int64)
    hw = (t[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64); xcc = (t[:, 3] >> np.uint64(32)).astype(np.int64) & 0xF
    tick = 10.0   # ns
    print(f"kernel span {(t0 + d_end).max() * tick / 1e3:.2f} us; waves {n}")
    print(f"per wave: load wait {d_load.mean() * tick / 1e3:.2f} us (p10 {np.percentile(d_load, 10) * tick / 1e3:.2f}, p90 {np.percentile(d_load, 90) * tick / 1e3:.2f}); "
          f"to stores {d_comp.mean() * tick / 1e3:.2f} us; lifetime {d_end.mean() * tick / 1e3:.2f} us (p10 {np.percentile(d_end, 10) * tick / 1e3:.2f}, p90 {np.percentile(d_end, 90) * tick / 1e3:.2f})")
    # occupancy timeline: resident waves per SIMD over time
    span = int((t0 + d_end).max()) + 1
    occ = np.zeros(span + 1, dtype=np.int64)
    np.add.at(occ, t0, 1); np.add.at(occ, t0 + d_end, -1)
    occ = np.cumsum(occ)[:span]
    comp = np.zeros(span + 1, dtype=np.int64)
    np.add.at(comp, t0 + d_load, 1); np.add.at(comp, t0 + d_comp, -1)
    comp = np.cumsum(comp)[:span]
    step = max(1, span // 40)
    print("time_us  resident_waves/SIMD  computing_waves/SIMD  starts_in_bin")
    starts = np.bincount(t0 // step, minlength=span // step + 1)
    for b in range(0, span, step):
        print(f"{b * tick / 1e3:7.2f}  {occ[b:b + step].mean() / 1024:6.2f}  {comp[b:b + step].mean() / 1024:6.2f}  {starts[b // step]:6d}")
    cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
    key = xcc * 10000 + se * 1000 + cu * 10 + simd
    u, cnt = np.unique(key, return_counts=True)
    print(f"distinct (xcc,se,cu,simd) = {len(u)}; waves per SIMD min {cnt.min()} max {cnt.max()} mean {cnt.mean():.1f}")
    print("waves per XCC:", np.bincount(xcc))
    print("waves-per-SIMD histogram:", dict(zip(*np.unique(cnt, return_counts=True))))
    cukey = xcc * 1000 + se * 100 + cu
    uc, ccnt = np.unique(cukey, return_counts=True)
    print(f"CUs {len(uc)}: waves per CU min {ccnt.min()} max {ccnt.max()}; histogram", dict(zip(*np.unique(ccnt, return_counts=True))))
    # when does each SIMD finish its last wave, and how does that relate to the number of waves it ran
    end = t0 + d_end
    inv = np.searchsorted(u, key)
    last = np.zeros(len(u), dtype=np.int64); np.maximum.at(last, inv, end)
    first_free = np.full(len(u), 1 << 60, dtype=np.int64)
    for c in sorted(set(cnt)):
        sel = cnt == c
        print(f"  SIMDs with {c:2d} waves: {sel.sum():4d}  last wave ends at {last[sel].mean() * tick / 1e3:6.2f} us (max {last[sel].max() * tick / 1e3:6.2f})")
    print("lifetime percentiles us:", {q: round(float(np.percentile(d_end, q)) * tick / 1e3, 2) for q in (1, 10, 50, 90, 99, 99.9, 100)})
    late = np.argsort(end)[-24:]
    print("last finishers: qidx start_us load_wait_us lifetime_us end_us")
    for i in late:
        print(f"   {i:6d} {t0[i] * tick / 1e3:7.2f} {d_load[i] * tick / 1e3:6.2f} {d_end[i] * tick / 1e3:6.2f} {end[i] * tick / 1e3:7.2f}")
    # lifetime of the waves that start late (after 15 us) vs early
    for lo, hi in ((0, 1), (5, 10), (10, 15), (15, 19), (19, 30)):
        sel = (t0 * tick / 1e3 >= lo) & (t0 * tick / 1e3 < hi)
        if sel.sum():
            print(f"  waves starting in [{lo},{hi}) us: {sel.sum():5d}  lifetime mean {d_end[sel].mean() * tick / 1e3:5.2f} p99 {np.percentile(d_end[sel], 99) * tick / 1e3:5.2f} max {d_end[sel].max() * tick / 1e3:5.2f}  load wait {d_load[sel].mean() * tick / 1e3:5.2f}")
    # per-XCC end time
    for x in range(8):
        print(f"  XCC {x}: last wave ends {end[xcc == x].max() * tick / 1e3:6.2f} us, first start {t0[xcc == x].min() * tick / 1e3:5.2f}, mean lifetime {d_end[xcc == x].mean() * tick / 1e3:5.2f}")
    # is the workgroup -> CU binding static?  per CU: when it finishes, and which workgroups it ran
    invc = np.searchsorted(uc, cukey)
    lastc = np.zeros(len(uc), dtype=np.int64); np.maximum.at(lastc, invc, end)
    print("per-CU last end us percentiles:", {q: round(float(np.percentile(lastc, q)) * tick / 1e3, 2) for q in (0, 10, 50, 90, 100)})
    sekey = xcc * 10 + se
    for k in np.unique(sekey):
        sel = sekey == k
        print(f"  XCC {k // 10} SE {k % 10}: CUs {len(np.unique(cukey[sel]))} waves {sel.sum()} last end {end[sel].max() * tick / 1e3:6.2f} mean lifetime {d_end[sel].mean() * tick / 1e3:5.2f}")
    wg = np.arange(n) // 2
    for c in (uc[0], uc[1], uc[len(uc) // 2], uc[np.argmax(lastc)], uc[np.argmin(lastc)]):
        sel = cukey == c
        ids = np.unique(wg[sel])
        print(f"  CU {c}: last end {lastc[np.searchsorted(uc, c)] * tick / 1e3:6.2f} us; workgroups ran (id // 8):", (ids // 8).tolist()[:40], "xcc of ids", np.unique(ids % 8).tolist())


if __name__ == "__main__":
    main()
