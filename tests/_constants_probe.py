"""Helper of tests/test_constants_override.py (run as a script in a fresh process, because both libraries are loaded once
per process): with JPEZY_LIB / JPEZY_ORACLE_LIB selecting a build, compare the product with the oracle on fixed inputs and
print one JSON line of digests and verdicts.  --gpu adds the HIP kernels (needs a device)."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes() if not isinstance(a, (bytes, bytearray)) else bytes(a))
    return h.hexdigest()[:16]


def main():
    gpu = "--gpu" in sys.argv
    from oracle import oracle as O
    import jpezy_amd as J
    out = {}
    # inputs that are sensitive to the unpinned constants: flat blocks of every level (DC = int(((8c * S) * S) / 4) / Q sits
    # on a truncation boundary for 255 of 256 levels), a frame of random pixels, odd sizes
    W, H = 256, 256
    lv = (np.arange(W * H) // (16 * W) * 16 + (np.arange(W * H) % W) // 16).astype(np.uint8)       # 16x16 flat patches 0..255
    flat = (lv, lv, lv)
    rnd = O.synth_rgb(W, H, frame=77)
    odd = O.synth_rgb(52, 40, frame=5)
    cases = {"flat": (flat, W, H), "rand": (rnd, W, H), "odd": (odd, 52, 40)}
    verdict = {}
    for name, ((r, g, b), w, h) in cases.items():
        for gray in (False, True):
            key = f"{name}{'_gray' if gray else ''}"
            co = O.encode_coeffs(r, g, b, w, h, gray)
            jpg = O.write_jpeg(co, w, h, gray)
            out[key + "_coeffs"] = sha(co)
            out[key + "_jpg"] = sha(jpg)
            verdict[key + "_host_writer"] = J.write_jpeg(co, w, h, gray) == jpg          # host Huffman/JFIF tail of the product
            info, back = J.read_jpeg(jpg)
            verdict[key + "_host_reader"] = bool(np.array_equal(back.reshape(-1), (co if not gray else O.read_jpeg(jpg)[1]).reshape(-1)))
            if gpu:
                from jpezy_amd import api as _api
                ctx = _api.default_context()
                got = ctx.fdct_quant(r, g, b, w, h, gray=gray)
                verdict[key + "_gpu_fdct"] = bool(np.array_equal(got, co))
                for force in (1, 2):
                    ctx.set_force_exact(force)
                    verdict[key + f"_gpu_fdct_force{force}"] = bool(np.array_equal(ctx.fdct_quant(r, g, b, w, h, gray=gray), co))
                ctx.set_force_exact(0)
                verdict[key + "_gpu_jpg"] = ctx.encode_jpeg(r, g, b, w, h, gray=gray) == jpg      # GPU entropy stage
                if not gray:
                    want = O.decode_planes(co, O.make_info(w, h), False)
                    gotp = ctx.dequant_idct(co, w, h)
                    verdict[key + "_gpu_idct"] = all(np.array_equal(a, e) for a, e in zip(gotp, want))
                    out[key + "_planes"] = sha(*want)
    out["verdict"] = verdict
    out["all_equal"] = all(verdict.values())
    out["constants"] = {"inv_sqrt2": float(O.constants()["inv_sqrt2"]).hex(), "cos9": float(O.constants()["cos"][9]).hex()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
