#!/usr/bin/env python3
"""Differential fuzz of jpezy_decode_jpeg_batch's batch form (round 3) against the per-file decode: groups of libjpeg files (4:2:0 = jpezy's
own layout, 4:4:4, 4:2:2, one component; random qualities / optimised tables / contents) and jpezy files of one size, with damaged copies mixed into the groups.
Every file must get the verdict and the planes the single-file call gives it.   python tools/fuzz/fuzz_decode_batch.py [groups] [seed]"""
import io
import sys
from pathlib import Path

import numpy as np
from PIL import Image, ImageFile

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests" / "fuzz"))
import jpezy_amd as J  # noqa: E402
from fuzz_huffdec import content  # noqa: E402
from run_host_fuzz import mutate  # noqa: E402

ImageFile.MAXBLOCK = 1 << 24


def single(ctx, data, gray):
    try:
        return ctx.decode_jpeg(data, gray=gray)
    except J.JpezyError:
        return None


def main():
    n_groups = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    ctx = J.Context(0)
    ctx.set_huffdec_min_bytes(0)
    n_files = n_fast = n_bad = 0
    for gi in range(n_groups):
        W, H = int(rng.integers(16, 700)), int(rng.integers(16, 500))
        files = []
        quality = int(rng.integers(10, 98))
        for k in range(int(rng.integers(3, 24))):
            img = content(rng, H, W, int(rng.integers(0, 5)))
            if rng.integers(0, 2):
                buf = io.BytesIO()
                sub = int(rng.choice([2, 2, 0, 1, 3]))                    # 4:2:0 (jpezy's own layout), 4:4:4, 4:2:2, one component
                if sub == 3:
                    Image.fromarray(img[..., 0]).save(buf, "JPEG", quality=quality, optimize=bool(rng.integers(0, 2)))
                else:
                    Image.fromarray(img).save(buf, "JPEG", quality=quality, subsampling=sub, optimize=bool(rng.integers(0, 2)))
                files.append(buf.getvalue())
            else:
                r, g, b = (np.ascontiguousarray(img[..., c]).reshape(-1) for c in range(3))
                files.append(ctx.encode_jpeg(r, g, b, W, H, gray=bool(rng.integers(0, 4) == 0)))
        for _ in range(int(rng.integers(0, 6))):                       # damaged copies somewhere in the group
            files.insert(int(rng.integers(0, len(files) + 1)), mutate(files[int(rng.integers(0, len(files)))], rng))
        gray = bool(rng.integers(0, 2))
        got = ctx.decode_jpeg_batch(files, gray=gray, raise_on_error=False)
        n_fast += ctx.last_batch_fast_count()
        for i, f in enumerate(files):
            want = single(ctx, f, gray)
            n_files += 1
            if (want is None) != (got[i] is None):
                print("VERDICT MISMATCH", gi, i, W, H, "single", "error" if want is None else "ok", "batch", "error" if got[i] is None else "ok")
                Path("gpurun_out").mkdir(exist_ok=True)
                Path(f"gpurun_out/batch_mutant_{gi}_{i}.jpg").write_bytes(f)
                return 1
            if want is None:
                n_bad += 1
                continue
            if want[0].width != got[i][0].width or any(not np.array_equal(a, b) for a, b in zip(want[1:], got[i][1:])):
                print("PLANES MISMATCH", gi, i, W, H)
                return 1
    print(f"fuzz_decode_batch: {n_groups} groups, {n_files} files ({n_bad} rejected by both paths), {n_fast} through the batch form of the kernels: "
          "every verdict and every plane equal to the single-file decode")
    return 0


if __name__ == "__main__":
    sys.exit(main())
