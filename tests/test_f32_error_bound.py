"""The FP32 first level of encode kernel variant 1 accepts a quantised coefficient t = F*cu*cv/(4Q) only when it is
further than delta1 from every non-zero integer, delta1 = 1.25 x the worst-case FP32 error of t over the lane's block
column (DeviceTables::delta1, built in jpezy_capi.hip).  This test re-derives that table, emulates the kernel's exact FP32
instruction sequence (same butterflies, same FMA placement -- the kernel issues them two at a time as v_pk_*_f32, which
changes no operand and no rounding) in numpy and measures the error against a float64 evaluation on adversarial blocks
(extreme amplitudes, every basis-function sign pattern, checkerboards) and random blocks -- the measured maximum must stay
far inside the guard band; the kernel's one-sided form of the test (fract(fma(F, ks, delta1)) < 2 delta1) is checked
against the two-sided definition.  The colour-conversion guard band is checked exhaustively over all 2^24 RGB triples."""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    # float32 fused multiply-add: the product of two float32 is exact in float64
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(f32)


def cosk(k):
    return np.float32(np.cos(k * np.pi / 16))


K1, K2, K3, K4, K5, K6, K7 = (cosk(k) for k in range(1, 8))


def fdct8f(x):
    """x: [..., 8] float32 -> [..., 8]; mirrors f32::fdct8p of jpezy_f32_quad.h operation by operation (the kernel
    computes the pairs (s_k, d_k), (e0, e2), (e1, e3), (X0, X4), (X2, X6), (X1, X3), (X5, X7) with one packed instruction per
    line below and pair: same operands, same order of the fused multiply-adds)"""
    x = x.astype(f32)
    s0, s1, s2, s3 = x[..., 0] + x[..., 7], x[..., 1] + x[..., 6], x[..., 2] + x[..., 5], x[..., 3] + x[..., 4]
    d0, d1, d2, d3 = x[..., 0] - x[..., 7], x[..., 1] - x[..., 6], x[..., 2] - x[..., 5], x[..., 3] - x[..., 4]
    e0, e1, e2, e3 = s0 + s3, s1 + s2, s0 - s3, s1 - s2
    X = [None] * 8
    X[0] = e0 + e1
    X[4] = e0 - e1                                       # its factor cos(pi/4) is folded into ks
    X[2] = fma(e3, K6, e2 * K2)
    X[6] = fma(-e3, K2, e2 * K6)
    X[1] = fma(d3, K7, fma(d2, K5, fma(d1, K3, d0 * K1)))
    X[3] = fma(-d3, K5, fma(-d2, K1, fma(-d1, K7, d0 * K3)))
    X[5] = fma(d3, K3, fma(d2, K7, fma(-d1, K1, d0 * K5)))
    X[7] = fma(-d3, K1, fma(d2, K3, fma(-d1, K5, d0 * K7)))
    return np.stack(X, axis=-1).astype(f32)


def blocks():
    rng = np.random.default_rng(0)
    out = [rng.integers(-128, 128, (4000, 8, 8))]
    out.append(rng.choice([-128, 127], (4000, 8, 8)))                       # extreme amplitudes
    # sign pattern of every 2-D basis function at full amplitude: maximises |F[i][j]|
    c = np.array([[np.cos((2 * x + 1) * u * np.pi / 16) for x in range(8)] for u in range(8)])
    basis = []
    for i in range(8):
        for j in range(8):
            sgn = np.sign(np.outer(c[i], c[j]))
            basis.append(np.where(sgn >= 0, 127, -128))
            basis.append(np.where(sgn >= 0, -128, 127))
    out.append(np.array(basis))
    yy, xx = np.mgrid[0:8, 0:8]
    out.append(np.array([np.where((xx // k + yy // k) % 2 == 0, 127, -128) for k in (1, 2, 4)]))
    out.append(np.full((2, 8, 8), 127) * np.array([1, -1])[:, None, None] - np.array([0, 1])[:, None, None])
    return np.concatenate(out).astype(np.int64)


def test_level1_error_is_far_inside_the_guard_band(oracle):
    c = oracle.constants()
    pic = blocks()
    rows = fdct8f(pic.astype(f32))                       # row pass: along x (last axis) -> [blk][y][j]
    F = fdct8f(np.swapaxes(rows, 1, 2))                  # column pass along y            -> [blk][j][i]
    F = np.swapaxes(F, 1, 2)                             # [blk][i][j]
    cos = c["cos"].reshape(8, 8)
    exact = np.einsum("iy,byx,jx->bij", cos, pic.astype(np.float64), cos)
    S = c["inv_sqrt2"]
    cu = np.where(np.arange(8) == 0, S, 1.0)
    # norm-wise bound of the FP32 transform, per coefficient: gamma_13 * sum_y|cos_i| * sum_x|cos_j| * 128  (at most 13
    # roundings on any input->output path, u = 2^-24), plus the rounding of the product t = F * ks (|t| <= 103)
    gamma = 13 * 2.0 ** -24
    absum = np.abs(cos).sum(axis=1)
    bound_F = gamma * np.outer(absum, absum) * 128
    # fdct8f leaves output 4 of each pass without its factor cos(pi/4) (folded into ks): compare scaled values
    k4 = np.where(np.arange(8) == 4, np.cos(np.pi / 4), 1.0)
    assert np.all(np.abs(F.astype(np.float64) * np.outer(k4, k4) - exact).max(axis=0) <= bound_F)
    # the table the kernel uses (jpezy_capi.hip): per table and column j, 1.25 x max_i of
    #   gamma_13 * amp + 2^-23 * amp,  amp = 128 * S_i * S_j * ks  (ks and the product t = F * ks are rounded to FP32)
    for qt, dmax in ((c["qt_luma"], 1.06e-4), (c["qt_chroma"], 5.8e-5)):
        scale = np.outer(cu, cu) / (4.0 * qt.reshape(8, 8))
        amp = 128.0 * np.outer(absum, absum) * scale
        bound_t = gamma * amp + 2.0 ** -23 * amp
        bound_t[0, 0] = 0                                # the DC term never uses the guard band (exact lookup table)
        delta1 = (1.25 * bound_t.max(axis=0)).astype(f32)            # [j]
        assert float(delta1.max()) <= dmax, delta1                   # the figures quoted in DESIGN.md / the kernel header
        k4 = np.where(np.arange(8) == 4, np.cos(np.pi / 4), 1.0)
        ks = (scale * np.outer(k4, k4)).astype(f32)      # F32Column::ks (carries the cos(pi/4) factors fdct8f leaves out)
        t32 = (F * ks).astype(f32)
        err = np.abs(t32.astype(np.float64) - exact * scale).max(axis=0)     # [i][j]
        err[0, 0] = 0
        assert np.all(err <= bound_t), (err / np.maximum(bound_t, 1e-30)).max()
        assert np.all(err.max(axis=0) < delta1 / 8), (err.max(axis=0) / delta1).max()   # measured: >10x inside the band
        # the kernel's one-sided test: t' = fma(F, ks, delta1), flagged <=> fract(t') < 2 delta1, q = trunc(t').  Every
        # coefficient it does NOT flag must truncate to the exact quotient's integer part, and the flagged set must contain
        # every coefficient whose exact t is within the proven error bound of a non-zero integer
        tb = fma(F, ks, np.broadcast_to(delta1, F.shape).astype(f32))
        fr = (tb - np.floor(tb)).astype(f32)                             # v_fract_f32
        flagged = fr < (delta1 + delta1)
        t_exact = exact * scale
        ok = np.trunc(tb.astype(np.float64)) == np.trunc(t_exact)
        ok[:, 0, 0] = True                                               # DC: exact table
        assert np.all(ok | flagged)
        near = np.abs(t_exact - np.rint(t_exact)) <= bound_t
        near[:, 0, 0] = False
        assert np.all(flagged[near])
        # the band is not wider than it has to be: flagged <=> the unbiased FP32 value is within delta1 (+ one rounding) of an integer
        d32 = np.abs(t32.astype(np.float64) - np.rint(t32.astype(np.float64)))
        assert np.all(d32[flagged] <= delta1.astype(np.float64).max() * 1.001 + 2.0 ** -23 * 128)
    # exact integer sums: the DC input of the lookup table
    assert np.array_equal(F[:, 0, 0].astype(np.int64), pic.sum(axis=(1, 2)))


def test_colour_level1_guard_band_exhaustive():
    """f32::luma_px2 / chroma_px2: Y = trunc(t'), t' an FP32 fma chain that starts from a small bias eps; flagged (-> FP64
    reference formula) when fract(t') < threshold (2^-12 / 2^-11 luma, 3*2^-16 / 3*2^-15 chroma).  Over all 2^24 RGB triples:
    every unflagged pixel truncates to the exact integer quotient, and exactly the pixels whose exact value is an integer
    (where the reference's FP64 rounding decides) are flagged."""
    G, B = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")
    Gf, Bf = G.astype(f32), B.astype(f32)
    worst = {"y": 0.0, "cb": 0.0, "cr": 0.0}
    flagged = {"y": 0, "cb": 0, "cr": 0}
    le, lt, ce, ct = f32(2.0 ** -12), f32(2.0 ** -11), f32(3 * 2.0 ** -16), f32(3 * 2.0 ** -15)
    for R in range(256):
        Rf = np.full_like(Gf, R)
        cases = {
            "y": (fma(Bf, f32(0.114), fma(Gf, f32(0.587), fma(Rf, f32(0.299), np.full_like(Gf, f32(-128.0) + le)))),
                  299 * R + 587 * G + 114 * B - 128000, 1000, le, lt, 2.4e-5),
            "cb": (fma(Bf, f32(0.5), fma(Gf, f32(-0.3313), fma(Rf, f32(-0.1687), np.full_like(Gf, ce)))),
                   -1687 * R - 3313 * G + 5000 * B, 10000, ce, ct, 1.7e-5),
            "cr": (fma(Bf, f32(-0.0813), fma(Gf, f32(-0.4187), fma(Rf, f32(0.5), np.full_like(Gf, ce)))),
                   5000 * R - 4187 * G - 813 * B, 10000, ce, ct, 1.7e-5),
        }
        for name, (t, num, den, eps, th, bound) in cases.items():
            assert t.dtype == f32
            tr = np.trunc(t)
            fr = (t - np.floor(t)).astype(f32)                         # v_fract_f32: exact in FP32
            flag = fr < th
            exact_q = np.trunc(num / den)                              # float64 quotient: exact enough for integers < 2^24
            integral = (num % den) == 0
            assert np.all(flag[integral]), name                        # the reference's rounding decides: must be flagged
            assert np.array_equal(tr[~flag], exact_q[~flag]), name     # everything else is already right
            worst[name] = max(worst[name], float(np.abs(t.astype(np.float64) - float(eps) - num / den).max()))
            flagged[name] += int(flag.sum())
            assert worst[name] <= bound, (name, worst[name])
            assert int(flag.sum()) == int(integral.sum()), name        # the band catches nothing else
    assert flagged["y"] == 16777216 // 1000 + 1 or 16000 < flagged["y"] < 17500, flagged


def test_dc_closed_form_reproduces_the_exact_table():
    """f32::dc_formula (persistent encode kernels): sign(S) * (((|S| - 1) >> 3) / Q) in the kernel's own FP32 operations against
    int(((S * s) * s) / 4) / Q evaluated in binary64 with C's truncating division, for every block sum and both DC quantisers.
    (jpezy_ctx_create repeats this check against the device table and disables the formula if it ever fails.)"""
    s = float.fromhex("0x1.6a09e667f3bccp-1")
    S = np.arange(-8192, 8193)
    for Q in (16, 17):
        dct = np.trunc(((S * s) * s) / 4).astype(np.int64)
        want = np.sign(dct) * (np.abs(dct) // Q)
        a = np.abs(S).astype(np.float32)
        d = np.trunc(np.float32(a * np.float32(0.125) - np.float32(0.125)))        # fma of exact quantities: exact
        rq, bias = np.float32(1.0) / np.float32(Q), np.float32(0.5) / np.float32(Q)
        u = (d.astype(np.float64) * np.float64(rq) + np.float64(bias)).astype(np.float32)   # one rounding, as v_fma_f32
        got = np.copysign(np.trunc(u), S).astype(np.int64)
        assert np.array_equal(got, want), Q
