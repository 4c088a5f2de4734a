"""The two numerical bounds the decode kernel's modes rest on (jpezy_amd/csrc/jpezy_kernels.hip, jpezy_capi.hip upload_dequant),
re-derived on the CPU with an emulation of the kernel's butterfly sequence:

* exact mode trusts its FP64 fast path for every int16 coefficient as long as |c * Q| <= 2^23 (every 8-bit quantiser table):
  the error of the two butterfly passes must stay far below the 2^-18 guard band even with all 64 inputs at that limit;
* tolerance mode runs the luma transforms in FP32 for |c * Q| <= 2^15: the error must stay below 1, so that a truncated sample
  differs from the reference's by at most one.
The emulation uses separate multiply and add where the kernel fuses them (an FMA rounds once, so the kernel errs less);
the yardstick is the same transform in x87 extended precision (64-bit mantissa)."""
import numpy as np

C = [np.cos(k * np.pi / 16) for k in range(8)]
S = 1.0 / np.sqrt(2.0)


def idct8(X, dt):
    """the kernel's even/odd butterflies (idct8 / idct8f), along axis 0, in dtype dt; X[0] already carries its 1/sqrt2"""
    c = [dt(v) for v in C]
    X = [X[k].astype(dt) for k in range(8)]
    t0 = X[4] * c[4] + X[0]; t1 = -X[4] * c[4] + X[0]
    t2 = X[6] * c[6] + X[2] * c[2]; t3 = -X[6] * c[2] + X[2] * c[6]
    E0, E3, E1, E2 = t0 + t2, t0 - t2, t1 + t3, t1 - t3
    O0 = X[7] * c[7] + (X[5] * c[5] + (X[3] * c[3] + X[1] * c[1]))
    O1 = -X[7] * c[5] + (-X[5] * c[1] + (-X[3] * c[7] + X[1] * c[3]))
    O2 = X[7] * c[3] + (X[5] * c[7] + (-X[3] * c[1] + X[1] * c[5]))
    O3 = -X[7] * c[1] + (X[5] * c[3] + (-X[3] * c[5] + X[1] * c[7]))
    return np.stack([E0 + O0, E1 + O1, E2 + O2, E3 + O3, E3 - O3, E2 - O2, E1 - O1, E0 - O0])


def samples(cq, dt):
    """cq: [n, 8 (v), 8 (u)] products coefficient * quantiser -> samples v = sum / 4 + 128 as the kernel forms them"""
    cucv = np.array([[(S if u == 0 else 1.0) * (S if v == 0 else 1.0) for u in range(8)] for v in range(8)])
    scale = (cucv * 0.25).astype(dt)                          # dqscale without Q (cq already carries it)
    x = (cq.astype(dt) * scale)                               # [n, v, u]
    col = idct8(np.moveaxis(x, 1, 0), dt)                     # over v -> [y, n, u]
    col = np.moveaxis(col, 0, 1)                              # [n, y, u]
    row_in = np.moveaxis(col, 2, 0).copy()                    # [u, n, y]
    row_in[0] = row_in[0] + dt(128.0)
    return np.moveaxis(idct8(row_in, dt), 0, 2)               # [n, y, x]


def _inputs(limit, rng, n=4000):
    blocks = [rng.integers(-limit, limit + 1, (n, 8, 8)).astype(np.float64)]
    signs = rng.integers(0, 2, (n, 8, 8)) * 2 - 1
    blocks.append(signs * float(limit))                       # every coefficient at the limit
    for u in range(8):                                        # the basis functions' own sign patterns: worst-case alignment
        for v in range(8):
            sx = np.sign(np.cos((2 * np.arange(8) + 1) * u * np.pi / 16) + 1e-30)
            sy = np.sign(np.cos((2 * np.arange(8) + 1) * v * np.pi / 16) + 1e-30)
            blocks.append((np.outer(sy, sx) * limit)[None])
    return np.concatenate(blocks)


def test_exact_mode_fast_path_error_is_far_below_the_guard_band():
    rng = np.random.default_rng(5)
    cq = _inputs(1 << 23, rng)
    got = samples(cq, np.float64)
    ref = samples(cq, np.longdouble)
    err = float(np.abs(got.astype(np.longdouble) - ref).max())
    bound = (8 * 6 * 8 + 6 * 64) * 2.0 ** 21 * 2.0 ** -53          # jpezy_capi.hip: 1.8e-7
    assert err < bound < 2.0 ** -18 / 20, (err, bound)
    assert float(np.abs(ref).max()) < 2.0 ** 28                    # samples still fit the int32 conversion


def test_tolerance_mode_fp32_luma_error_is_below_one():
    rng = np.random.default_rng(6)
    cq = _inputs(1 << 15, rng)
    got = samples(cq, np.float32)
    ref = samples(cq, np.longdouble)
    err = float(np.abs(got.astype(np.longdouble) - ref).max())
    assert err < (13 + 2) * 2.0 ** -24 * 2.0 ** 19 < 1.0, err      # the bound quoted in jpezy_kernels.hip: 0.47
    # and therefore truncated samples differ by at most one
    assert int(np.abs(np.trunc(got.astype(np.float64)) - np.trunc(ref.astype(np.float64))).max()) <= 1
