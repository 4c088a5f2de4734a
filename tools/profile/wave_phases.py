#!/usr/bin/env python3
"""Development probe: where a wave of the f32 encode kernel spends its life.  Build with
    python tools/ab/ab_build.py phases:-DJPEZY_TRACE=3
and run with JPEZY_LIB=ab/libjpezy_phases.so.  Every wave stamps the shader clock (s_memtime) at entry (0), when its pixels
are in registers (1), after the luma estimate + row pass (2), after the chroma estimate (3), after the column reads (4),
after the luma quantiser (5), after the chroma transforms + quantiser (6), after levels 2/3 (7) and at its end (8)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402
from jpezy_amd import api  # noqa: E402

NAMES = ["wait for pixels", "luma estimate + row pass", "chroma estimate", "tile sync + column reads", "luma column pass + quantiser",
         "chroma passes + quantiser", "levels 2/3", "stores issued"]


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
    ph = buf[4 * 65536:4 * 65536 + n * 9].reshape(n, 9).astype(np.int64)
    t0 = t[:, 0].astype(np.int64); t0 -= t0.min()
    d_end = t[:, 2].astype(np.int64)                       # 10 ns ticks
    d = np.diff(ph, axis=1)                                # [wave][8] cycles
    life = ph[:, 8] - ph[:, 0]
    ghz = np.median(life / (d_end * 10.0))
    print(f"kernel span {(t0 + d_end).max() / 100:.2f} us; shader clock {ghz:.2f} GHz (median of cycles / ns over the waves)")
    order = np.argsort(t0)
    groups = {"all waves": np.arange(n), "first 5000 started (ramp)": order[:5000], "middle (steady state)": order[7000:12000],
              "last 1000 started (drain)": order[-1000:], "last 200 started": order[-200:]}
    for name, idx in groups.items():
        print(f"== {name}: lifetime {life[idx].mean() / ghz / 1e3:.2f} us mean, {np.median(life[idx]) / ghz / 1e3:.2f} median")
        for k in range(8):
            x = d[idx, k]
            print(f"   {NAMES[k]:32s} mean {x.mean():8.0f} cyc  median {np.median(x):8.0f}  p90 {np.percentile(x, 90):8.0f}  max {x.max():8.0f}   ({x.mean() / life[idx].mean() * 100:4.1f} %)")


if __name__ == "__main__":
    main()
