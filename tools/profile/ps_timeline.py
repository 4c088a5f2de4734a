#!/usr/bin/env python3
"""Development probe for the persistent encode kernels (variant 3: quad q of a 4096^2 frame belongs to wave q % 4096, round q // 4096):
per-round start / end times of the waves, gaps between a wave's rounds, per-XCD spread.  Build: tools/ab/ab_build.py
phases_ps2:"-DJPEZY_DEFAULT_VARIANT=3 -DJPEZY_TRACE=3"; run with JPEZY_LIB=ab/libjpezy_phases_ps2.so."""
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
    buf = np.zeros(13 * 65536, dtype=np.uint64)
    lib.jpezy_debug_read_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.jpezy_debug_read_trace(ctx._h, buf.ctypes.data, buf.size)
    t = buf[:n * 4].reshape(n, 4)
    t0 = t[:, 0].astype(np.int64); base = t0.min(); t0 -= base          # 10 ns ticks
    d_end = t[:, 2].astype(np.int64)
    xcc = (t[:, 3] >> np.uint64(32)).astype(np.int64) & 0xF
    nw = 4096
    start = t0.reshape(4, nw) / 100.0       # us, [round][wave]
    end = (t0 + d_end).reshape(4, nw) / 100.0
    print(f"kernel span (first quad start -> last quad end): {end.max():.2f} us")
    for r in range(4):
        print(f"round {r}: start mean {start[r].mean():6.2f} (p10 {np.percentile(start[r], 10):6.2f} p90 {np.percentile(start[r], 90):6.2f})   "
              f"end mean {end[r].mean():6.2f} (p90 {np.percentile(end[r], 90):6.2f} max {end[r].max():6.2f})   length mean {(end[r] - start[r]).mean():5.2f}")
    for r in range(3):
        gap = start[r + 1] - end[r]
        print(f"gap between round {r} and {r + 1}: mean {gap.mean():5.2f} us  p90 {np.percentile(gap, 90):5.2f}  max {gap.max():5.2f}")
    fin = end[3]
    x = xcc.reshape(4, nw)[0]
    for k in range(8):
        m = x == k
        if m.any():
            print(f"XCC {k}: {m.sum():5d} waves, first start {start[0][m].min():5.2f}, finish mean {fin[m].mean():6.2f} max {fin[m].max():6.2f}")
    wg_fin = fin.reshape(256, 16).max(axis=1)
    print(f"workgroup finish: mean {wg_fin.mean():.2f}  p10 {np.percentile(wg_fin, 10):.2f}  p90 {np.percentile(wg_fin, 90):.2f}  max {wg_fin.max():.2f}")


if __name__ == "__main__":
    main()
