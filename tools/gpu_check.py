#!/usr/bin/env python3
"""Quick GPU parity probe (development aid; the judged tests live in tests/)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import oracle as O  # noqa: E402
import jpezy_amd as J  # noqa: E402


def main():
    ctx = J.Context(0)
    ok = True
    cases = [(16, 16, False), (64, 48, False), (33, 17, False), (17, 33, True), (160, 96, True), (512, 512, False),
             (1920, 1080, False)]
    for (W, H, gray) in cases:
        r, g, b = O.synth_rgb(W, H)
        want = O.encode_coeffs(r, g, b, W, H, gray)
        for force in (False, True):
            if force and W * H > 64 * 64:
                continue
            ctx.set_force_exact(force)
            t = time.time()
            got = ctx.fdct_quant(r, g, b, W, H, gray)
            dt = time.time() - t
            nfb = ctx.fallback_count()
            bad = int((got != want).sum())
            print(f"enc {W}x{H} gray={gray} force={force}: mismatches={bad} fallbacks={nfb} ({dt*1e3:.1f} ms)")
            if bad:
                ok = False
                idx = np.argwhere(got != want)[:8]
                for i in idx:
                    print("   at", tuple(i), "got", got[tuple(i)], "want", want[tuple(i)])
        ctx.set_force_exact(False)
        # decode
        co6 = O.encode_coeffs(r, g, b, W, H, False)
        info = O.make_info(W, H)
        for dgray in (False, True):
            wr, wg, wb = O.decode_planes(co6, info, dgray)
            for force in (False, True):
                if force and W * H > 64 * 64:
                    continue
                ctx.set_force_exact(force)
                gr, gg, gb = ctx.dequant_idct(co6, W, H, gray=dgray)
                nfb = ctx.fallback_count()
                bad = int((gr != wr).sum() + (gg != wg).sum() + (gb != wb).sum())
                mx = max(int(np.abs(gr.astype(int) - wr).max()), int(np.abs(gg.astype(int) - wg).max()),
                         int(np.abs(gb.astype(int) - wb).max()))
                print(f"dec {W}x{H} gray={dgray} force={force}: mismatches={bad} maxdiff={mx} fallbacks={nfb}")
                if bad:
                    ok = False
            ctx.set_force_exact(False)
    print("ALL OK" if ok else "FAILURES")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
