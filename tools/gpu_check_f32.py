#!/usr/bin/env python3
"""Development probe for encode kernel variant 1 (FP32 first level)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import oracle as O  # noqa: E402
import jpezy_amd as J  # noqa: E402


def main():
    ctx = J.Context(0)
    ctx.set_variant(1)
    ok = True
    cases = [(16, 16, False), (64, 48, False), (33, 17, False), (17, 33, True), (160, 96, True), (512, 512, False),
             (1920, 1080, False), (4096, 4096, False)]
    for (W, H, gray) in cases:
        r, g, b = O.synth_rgb(W, H, frame=W + H)
        want = O.encode_coeffs(r, g, b, W, H, gray)
        for force in (0, 2, 1):
            if force and W * H > 64 * 64:
                continue
            ctx.set_force_exact(force)
            t = time.time()
            got = ctx.fdct_quant(r, g, b, W, H, gray)
            dt = time.time() - t
            nfb = ctx.fallback_count()
            bad = int((got != want).sum())
            print(f"enc {W}x{H} gray={gray} force={force}: mismatches={bad} level2+3={nfb} ({dt*1e3:.1f} ms)", flush=True)
            if bad:
                ok = False
                for i in np.argwhere(got != want)[:6]:
                    print("   at", tuple(i), "got", got[tuple(i)], "want", want[tuple(i)])
        ctx.set_force_exact(0)
    print("ALL OK" if ok else "FAILURES")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
