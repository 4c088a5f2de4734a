#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/jpezy_oracle.c) in this container.

PARITY UNPINNED: the reference (falgon/jpezy) holds no test vectors and cannot be built here (its
SrookCppLibraries / Boost dependencies are absent), so these fixtures freeze the ORACLE's outputs -- they
pin the oracle and the HIP path against regressions and against each other, not against a reference run.
Each fixture: input planes, zig-zag coefficients (colour and gray), .jpg bytes (colour and gray), decoded
planes (colour and gray).  Larger cases are pinned by SHA-256 only (tests/golden/digests.json).
"""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import oracle as O  # noqa: E402

OUT = ROOT / "tests" / "golden"


def images():
    yield "rand16", 16, 16, O.synth_rgb(16, 16, frame=1)
    yield "rand17x33", 17, 33, O.synth_rgb(17, 33, frame=2)          # edge clamp in both directions
    yield "rand64", 64, 64, O.synth_rgb(64, 64, frame=3)
    # flat greys 0..255: one 16x16 MCU per level, 16 MCUs per row -> the systematic DC = 8c-/+1 case (H3)
    lv = np.repeat(np.repeat(np.arange(256, dtype=np.uint8).reshape(16, 16), 16, axis=0), 16, axis=1).reshape(-1)
    yield "flatgrey256", 256, 256, (lv.copy(), lv.copy(), lv.copy())
    # grey ramp: hits the 30 grey levels whose Y differs from c-128 (H2)
    ramp = np.tile(np.arange(256, dtype=np.uint8), 16)
    yield "greyramp256x16", 256, 16, (ramp.copy(), ramp.copy(), ramp.copy())
    # smooth colour gradients, odd size
    yy, xx = np.mgrid[0:40, 0:52]
    yield "gradient52x40", 52, 40, ((xx * 5 % 256).astype(np.uint8).reshape(-1), (yy * 6 % 256).astype(np.uint8).reshape(-1),
                                    ((xx + yy) * 3 % 256).astype(np.uint8).reshape(-1))


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    for name, W, H, (r, g, b) in images():
        co = O.encode_coeffs(r, g, b, W, H, False)
        cog = O.encode_coeffs(r, g, b, W, H, True)
        jpg = O.write_jpeg(co, W, H, False)
        jpgg = O.write_jpeg(cog, W, H, True)
        info = O.make_info(W, H)
        dr, dg, db = O.decode_planes(co, info, False)
        gr, _, _ = O.decode_planes(co, info, True)
        np.savez_compressed(OUT / f"{name}.npz", W=W, H=H, r=r, g=g, b=b, coeffs=co, coeffs_gray=cog,
                            jpg=np.frombuffer(jpg, np.uint8), jpg_gray=np.frombuffer(jpgg, np.uint8),
                            dec_r=dr, dec_g=dg, dec_b=db, dec_gray=gr)
        print(name, W, H, len(jpg), len(jpgg))
    digests = {}
    for name, W, H, frame in [("rand512", 512, 512, 0), ("rand1920x1080", 1920, 1080, 0), ("rand720x486", 720, 486, 5)]:
        r, g, b = O.synth_rgb(W, H, frame=frame)
        co = O.encode_coeffs(r, g, b, W, H, False)
        jpg = O.write_jpeg(co, W, H, False)
        dr, dg, db = O.decode_planes(co, O.make_info(W, H), False)
        digests[name] = {"W": W, "H": H, "frame": frame,
                         "coeffs_sha256": hashlib.sha256(co.tobytes()).hexdigest(),
                         "jpg_sha256": hashlib.sha256(jpg).hexdigest(), "jpg_len": len(jpg),
                         "dec_sha256": hashlib.sha256(dr.tobytes() + dg.tobytes() + db.tobytes()).hexdigest()}
        print(name, digests[name]["jpg_len"])
    (OUT / "digests.json").write_text(json.dumps(digests, indent=1) + "\n")


if __name__ == "__main__":
    main()
