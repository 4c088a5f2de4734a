#!/usr/bin/env python3
"""Soak test (run by hand on the GPU box, not collected by pytest):

    python tests/soak_parity.py [seconds] [workers] [seed]

For `seconds` of wall time: random frames of many kinds of content and sizes go through the HIP encode and decode
kernels (C-ABI, device-pointer entry points) and through the CPU oracle (worker processes); every coefficient and every
decoded byte must be identical.  The point is statistics: the fast paths reproduce a truncating FP64 reference through
guard bands and exact fallbacks, so a wrong bound would show up as a rare mismatch -- this runs
10^10..10^11 samples.  Prints one summary line; exit code 1 on any mismatch (the failing case is saved under
gpurun_out/soak_fail_*.npz).
"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def make_frame(kind, W, H, rng):
    n = W * H
    if kind == "uniform":
        return [rng.integers(0, 256, n, dtype=np.uint8) for _ in range(3)]
    if kind == "lownoise":                       # small noise around a random level: samples crowd the truncation boundaries
        base = rng.integers(0, 256, 3)
        amp = int(rng.integers(1, 6))
        return [np.clip(base[c] + rng.integers(-amp, amp + 1, n), 0, 255).astype(np.uint8) for c in range(3)]
    if kind == "gradient":
        yy, xx = np.mgrid[0:H, 0:W]
        a, b, c = rng.integers(1, 9, 3)
        return [((xx * a + yy * b) // c % 256).astype(np.uint8).reshape(-1), ((xx * b + yy * c) // a % 256).astype(np.uint8).reshape(-1),
                ((xx * c + yy * a) // b % 256).astype(np.uint8).reshape(-1)]
    if kind == "flatblocks":                     # flat 8x8 / 16x16 patches: DC-only blocks, exact-integer samples
        s = int(rng.choice([8, 16, 32]))
        out = []
        for _ in range(3):
            small = rng.integers(0, 256, ((H + s - 1) // s, (W + s - 1) // s), dtype=np.uint8)
            out.append(np.repeat(np.repeat(small, s, axis=0), s, axis=1)[:H, :W].reshape(-1).copy())
        return out
    if kind == "binary":                         # saturated 0 / 255 noise: the largest coefficients
        return [(rng.integers(0, 2, n, dtype=np.uint8) * 255) for _ in range(3)]
    if kind == "grey":                           # r = g = b
        g = rng.integers(0, 256, n, dtype=np.uint8)
        return [g, g.copy(), g.copy()]
    if kind == "mult8":                          # values on a coarse grid: products and sums land on round numbers
        return [(rng.integers(0, 32, n, dtype=np.uint8) * 8) for _ in range(3)]
    if kind == "checker":
        yy, xx = np.mgrid[0:H, 0:W]
        p = int(rng.integers(1, 5))
        m = (((xx // p) + (yy // p)) & 1).astype(np.uint8).reshape(-1)
        lo, hi = sorted(int(v) for v in rng.integers(0, 256, 2))
        return [(lo + m * (hi - lo)).astype(np.uint8), (hi - m * (hi - lo)).astype(np.uint8), (lo + m * (hi - lo)).astype(np.uint8)]
    raise ValueError(kind)


KINDS = ["uniform", "lownoise", "gradient", "flatblocks", "binary", "grey", "mult8", "checker", "coeffs", "coeffs", "generic", "generic"]
SIZES = [(4096, 4096), (1920, 1080), (1237, 911), (640, 480), (4096, 2160), (333, 2047)]


def make_coeffs(W, H, rng):
    """decode-only case: coefficients no encoder of ours produced, with random quantiser tables"""
    mc, mr = (W + 15) // 16, (H + 15) // 16
    shape = (mc * mr * 6, 64)
    mode = int(rng.integers(0, 5))
    if mode == 0:      # dense, small
        co = rng.integers(-20, 21, shape, dtype=np.int16)
    elif mode == 1:    # sparse, large
        co = np.zeros(shape, np.int16)
        m = rng.random(shape) < 0.06
        co[m] = rng.integers(-2047, 2048, int(m.sum()), dtype=np.int16)
    elif mode == 2:    # whole int16 range in a few blocks: the magnitude guard of the fast path
        co = rng.integers(-60, 61, shape, dtype=np.int16)
        rows = rng.integers(0, shape[0], max(1, shape[0] // 50))
        co[rows] = rng.integers(-32768, 32768, (len(rows), 64), dtype=np.int16)
    elif mode == 3:    # DC only
        co = np.zeros(shape, np.int16)
        co[:, 0] = rng.integers(-1024, 1024, shape[0], dtype=np.int16)
    else:              # low frequencies only (what a real photograph looks like)
        co = np.zeros(shape, np.int16)
        co[:, :10] = rng.integers(-90, 91, (shape[0], 10), dtype=np.int16)
    qmax = int(rng.choice([8, 40, 255, 2000]))
    qt = rng.integers(1, qmax + 1, (3, 64))
    tq = [int(v) for v in rng.integers(0, 3, 3)]
    return co.reshape(-1), qt, tq


LAYOUTS = [[(1, 1), (1, 1), (1, 1)], [(2, 1), (1, 1), (1, 1)], [(2, 2), (1, 1), (1, 1)], [(1, 1)], [(2, 2)],
           [(4, 1), (1, 1), (1, 1)], [(1, 2), (1, 1), (1, 1)], [(4, 2), (2, 1), (2, 2)], [(3, 1), (1, 1), (2, 1)]]


def make_generic(W, H, rng):
    """any-layout decode case: layout, quantiser tables, selectors, coefficients"""
    lay = LAYOUTS[int(rng.integers(0, len(LAYOUTS)))]
    hmax, vmax = max(h for h, _ in lay), max(v for _, v in lay)
    hb, vb = (W + 7) // 8, (H + 7) // 8
    mc, mr = -(-hb // hmax), -(-vb // vmax)
    bpm = sum(h * v for h, v in lay)
    shape = (mc * mr * bpm, 64)
    mode = int(rng.integers(0, 4))
    if mode == 0:
        co = rng.integers(-25, 26, shape, dtype=np.int16)
    elif mode == 1:
        co = np.zeros(shape, np.int16)
        m = rng.random(shape) < 0.08
        co[m] = rng.integers(-1500, 1501, int(m.sum()), dtype=np.int16)
    elif mode == 2:
        co = np.zeros(shape, np.int16)
        co[:, 0] = rng.integers(-900, 900, shape[0], dtype=np.int16)
        co[:, 1:6] = rng.integers(-40, 41, (shape[0], 5), dtype=np.int16)
    else:
        co = rng.integers(-70, 71, shape, dtype=np.int16)
        rows = rng.integers(0, shape[0], max(1, shape[0] // 40))
        co[rows] = rng.integers(-32768, 32768, (len(rows), 64), dtype=np.int16)
    qt = rng.integers(1, int(rng.choice([10, 60, 255])) + 1, (3, 64))
    tq = [int(v) for v in rng.integers(0, 3, 3)]
    return lay, (hmax, vmax, mc, mr, bpm), co.reshape(-1), qt, tq


def fill_info(info, W, H, lay, geo, qt, tq):
    hmax, vmax, mc, mr, bpm = geo
    info.width, info.height, info.ncomp, info.precision = W, H, len(lay), 8
    for i, (h, v) in enumerate(lay):
        info.H[i], info.V[i], info.Tq[i] = h, v, tq[i]
    info.hmax, info.vmax, info.mcu_cols, info.mcu_rows, info.blocks_per_mcu = hmax, vmax, mc, mr, bpm
    for t in range(3):
        for i in range(64):
            info.qt[t][i] = int(qt[t][i])
    return info


def oracle_job(args):
    kind, W, H, seed, gray = args
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    if kind == "generic":
        lay, geo, co, qt, tq = make_generic(W, H, rng)
        info = fill_info(O.FrameInfo(), W, H, lay, geo, qt, tq)
        planes = O.decode_planes(co, info, gray)
        return None, [np.asarray(p).reshape(-1)[: W * H] for p in planes]
    if kind == "coeffs":
        co, qt, tq = make_coeffs(W, H, rng)
        info = O.make_info(W, H, gray_layout=False)
        for t in range(3):
            for i in range(64):
                info.qt[t][i] = int(qt[t][i])
        for i in range(3):
            info.Tq[i] = tq[i]
        planes = O.decode_planes(co, info, gray)
        return None, [np.asarray(p).reshape(-1)[: W * H] for p in planes]
    r, g, b = make_frame(kind, W, H, rng)
    co = O.encode_coeffs(r, g, b, W, H, gray=gray)
    info = O.make_info(W, H, gray_layout=False)
    co6 = co if not gray else O.encode_coeffs(r, g, b, W, H, gray=False)
    planes = O.decode_planes(co6, info, gray)
    return np.asarray(co).reshape(-1), [np.asarray(p).reshape(-1)[: W * H] for p in planes]


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, (os.cpu_count() or 4) - 2)
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else int(time.time()))
    t_end = time.time() + seconds
    done = px = bad = 0
    last = time.time()
    # the oracle workers are forked BEFORE this process touches the GPU (the first 2 * workers submissions start them all)
    with ProcessPoolExecutor(workers) as pool:
        pending = []
        case = 0

        def submit():
            nonlocal case
            kind = KINDS[case % len(KINDS)]
            W, H = SIZES[int(rng.integers(0, len(SIZES)))]
            if rng.random() < 0.3:
                W, H = int(rng.integers(1, 700)), int(rng.integers(1, 700))
            if kind == "generic" and W * H > 2_500_000:
                W, H = 1237, 911
            args = (kind, W, H, int(rng.integers(0, 2**31)), bool(rng.integers(0, 2)))
            pending.append((args, pool.submit(oracle_job, args)))
            case += 1

        for _ in range(workers * 2):
            submit()
        import torch
        import jpezy_amd as J
        ctx = J.Context(0)
        if os.environ.get("JPEZY_SOAK_VARIANT"):      # encode kernel variant under test (default: the library's default, 1)
            ctx.set_variant(int(os.environ["JPEZY_SOAK_VARIANT"]))
        dev = torch.device("cuda:0")
        case_no = -1
        while pending:
            case_no += 1
            args, fut = pending.pop(0)
            kind, W, H, seed, gray = args
            want_co, want_planes = fut.result()
            if time.time() < t_end:
                submit()
            if kind == "generic":
                lay, geo, co_h, qt, tq = make_generic(W, H, np.random.default_rng(seed))
                ginfo = fill_info(J.FrameInfo(), W, H, lay, geo, qt, tq)
                got = ctx.dequant_idct_generic(co_h, ginfo, gray=gray)
                ok = all(np.array_equal(o, w) for o, w in zip(got, want_planes))
                done += 1
                px += W * H
                if not ok:
                    bad += 1
                    print(f"MISMATCH: kind={kind} {W}x{H} seed={seed} gray={gray} layout={lay}", flush=True)
                continue
            if kind == "coeffs":
                co_h, qt, tq = make_coeffs(W, H, np.random.default_rng(seed))
                qtab = type(J.api.annex_k_tables().qt)()
                for t in range(3):
                    for i in range(64):
                        qtab[t][i] = int(qt[t][i])
                out = [torch.empty(W * H, dtype=torch.uint8, device=dev) for _ in range(3)]
                ctx.dequant_idct_dev(torch.from_numpy(co_h).to(dev), W, H, out[0], out[1], out[2], qt=qtab, comp_tq=tuple(tq), gray=gray)
                torch.cuda.synchronize()
                ok = all(np.array_equal(o.cpu().numpy(), w) for o, w in zip(out, want_planes))
                done += 1
                px += W * H
                if not ok:
                    bad += 1
                    print(f"MISMATCH: kind={kind} {W}x{H} seed={seed} gray={gray}", flush=True)
                continue
            r, g, b = make_frame(kind, W, H, np.random.default_rng(seed))
            # Every third case goes through the batch form of the entry points: the frame is copy `slot` of `nf` frames at a
            # plane stride that is not W*H, from a base address that is not 16-byte aligned (the unaligned kernel variants,
            # the frame index of the launch); the other frames of the batch hold noise.
            lrng = np.random.default_rng(seed ^ 0x5A5A)
            batched = case_no % 3 == 2 and W * H <= (1 << 22)
            nf = int(lrng.integers(2, 5)) if batched else 1
            slot = int(lrng.integers(0, nf)) if batched else 0
            pad = int(lrng.integers(1, 40)) if batched else 0
            skew = int(lrng.integers(1, 16)) if batched else 0
            stride = W * H + pad
            d = []
            for p in (r, g, b):
                buf = torch.randint(0, 256, (skew + nf * stride,), dtype=torch.uint8, device=dev)
                buf[skew + slot * stride: skew + slot * stride + W * H] = torch.from_numpy(p).to(dev)
                d.append(buf[skew:])
            nco = J.coeff_count(W, H, gray)
            co_all = torch.empty(nf * nco, dtype=torch.int16, device=dev)
            ctx.fdct_quant_dev(d[0], d[1], d[2], W, H, co_all, gray=gray, n_frames=nf, plane_stride=stride)
            co = co_all[slot * nco:(slot + 1) * nco]
            co6_all = co_all
            nco6 = J.coeff_count(W, H, False)
            if gray:
                co6_all = torch.empty(nf * nco6, dtype=torch.int16, device=dev)
                ctx.fdct_quant_dev(d[0], d[1], d[2], W, H, co6_all, gray=False, n_frames=nf, plane_stride=stride)
            obuf = [torch.zeros(skew + nf * stride, dtype=torch.uint8, device=dev) for _ in range(3)]
            ctx.dequant_idct_dev(co6_all, W, H, obuf[0][skew:], obuf[1][skew:], obuf[2][skew:], gray=gray, n_frames=nf, plane_stride=stride)
            out = [o[skew + slot * stride: skew + slot * stride + W * H] for o in obuf]
            torch.cuda.synchronize()
            ok = np.array_equal(co.cpu().numpy(), want_co) and all(np.array_equal(o.cpu().numpy(), w) for o, w in zip(out, want_planes))
            if batched and pad:      # nothing may be written between the frames
                ok = ok and all(int(o[skew + slot * stride + W * H: skew + (slot + 1) * stride].max()) == 0 for o in obuf)
            done += 1
            px += W * H
            if not ok:
                bad += 1
                outdir = ROOT / "gpurun_out"
                outdir.mkdir(exist_ok=True)
                np.savez_compressed(outdir / f"soak_fail_{done}.npz", kind=kind, W=W, H=H, seed=seed, gray=gray)
                print(f"MISMATCH: kind={kind} {W}x{H} seed={seed} gray={gray}", flush=True)
            if time.time() - last > 30:
                last = time.time()
                print(f"... {done} frames, {px / 1e9:.2f} Gpx, {bad} mismatches", flush=True)
    print(f"soak: {done} frames, {px / 1e9:.2f} Gpx encoded and decoded on the GPU and by the oracle ({workers} worker processes), "
          f"{bad} mismatches, exact-path samples on the GPU: {ctx.fallback_count()}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
