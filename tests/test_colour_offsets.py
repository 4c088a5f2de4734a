"""The decode kernel's colour conversion without FP64 (jpezy_amd/csrc/jpezy_kernels.hip, step 5): proof by enumeration.

The reference converts in doubles and truncates (ref decoder/jpezy_decoder.hpp:567-578, revise_value :672-676):
    r = Y + V * 1.4020,  g = Y - U * 0.3441 - V * 0.7139,  b = Y + U * 1.7718      (U = Cb - 128, V = Cr - 128, integers)
    plane value = r < 0 ? 0 : r > 255 ? 255 : (uint8_t)r
Y, U, V are integers, so the three values are Y plus a term that depends on the chroma sample alone, and
    clamp(trunc(Y + t)) == clamp(Y + floor(t))      (they differ only for negative values, which both clamp to 0)
whenever the double evaluation cannot land on the other side of an integer -- i.e. whenever t is not an integer itself:
1.402 V is a multiple of 0.002, 1.7718 U of 0.0002, 0.3441 U + 0.7139 V of 0.0001, and the reference's rounding errors
are ~1e-13.  The kernel therefore forms three integer offsets per chroma sample with FP32 arithmetic
    offR = floor(1.402f * V)   offB = floor(1.7718f * U)   c = fma(0.3441f, U, 0.7139f * V), offG = floor(-c)
adds them to the (two) luma samples as integers, and flags the chroma samples whose c is within DEC_CHROMA_BAND of a
non-zero integer: exactly those with 3441 U + 7139 V a non-zero multiple of 10000, where the reference's own rounding
sequence decides (1e-4 of all pairs; a wave with such a sample, or with |U|, |V| > DEC_CHROMA_GATE, converts in doubles
as before).  U = V = 0 -- every gray pixel -- has c = 0 exactly and is exempt: the reference subtracts two zeros.

This test emulates the kernel's FP32 operations bit for bit (products of two floats and the fused sum are exact in
doubles before the single rounding) for every pair inside the gate and checks all of the above against integer
arithmetic and against the reference's double formula."""
import numpy as np

GATE = 512                      # DEC_CHROMA_GATE
BAND = np.float32(5e-5)         # DEC_CHROMA_BAND
F = np.float32
K_R, K_G1, K_G2, K_B = F(1.402), F(0.3441), F(0.7139), F(1.7718)


def f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32)


def kernel_offsets(U, V):
    """the kernel's per-chroma-sample arithmetic; U, V integer arrays"""
    Uf, Vf = U.astype(np.float32), V.astype(np.float32)
    offR = np.floor(f32(np.float64(K_R) * Vf)).astype(np.int64)              # v_mul_f32, v_cvt_flr_i32_f32
    offB = np.floor(f32(np.float64(K_B) * Uf)).astype(np.int64)
    p = f32(np.float64(K_G2) * Vf)                                            # v_mul_f32
    c = f32(np.float64(K_G1) * Uf + np.float64(p))                            # v_fma_f32 (the sum is exact in doubles: < 40 bits)
    offG = np.floor(-c).astype(np.int64)                                      # v_cvt_flr_i32_f32 with the neg modifier
    n = np.rint(c)                                                            # v_rndne_f32
    d = (c - n).astype(np.float32)                                            # exact
    key = np.maximum(np.abs(d), F(1.0) - np.abs(n))                           # v_sub_f32 1.0, |n| ; v_max_f32 |d|, s
    return offR, offG, offB, key <= BAND


def test_offsets_equal_the_integer_floors_and_the_flags_are_the_integral_chroma_terms():
    u = np.arange(-GATE, GATE + 1, dtype=np.int64)
    U, V = np.meshgrid(u, u, indexing="ij")
    offR, offG, offB, flag = kernel_offsets(U, V)
    assert np.array_equal(offR, (701 * V) // 500)                             # floor(1.402 V)
    # 1.402 V is integral at V = 0 and, inside the gate, at V = +-500 (= +-701): there the argument "floor of a non-integer" does
    # not apply, and the offsets are right only because the FP32 product AND the reference's double product are both exactly 701
    # (made explicit so that a change of constant or gate that breaks either equality fails here)
    assert GATE >= 500 and float(f32(np.float64(K_R) * F(500.0))) == 701.0 and float(f32(np.float64(K_R) * F(-500.0))) == -701.0
    assert 500.0 * 1.4020 == 701.0 and -500.0 * 1.4020 == -701.0
    assert [int(v) for v in u if (701 * int(v)) % 500 == 0] == [-500, 0, 500]
    assert [int(v) for v in u if (8859 * int(v)) % 5000 == 0] == [0]          # 1.7718 U: integral only at zero inside the gate
    assert np.array_equal(offB, (8859 * U) // 5000)
    N = 3441 * U + 7139 * V
    integral = (N % 10000 == 0) & (N != 0)
    assert np.array_equal(flag, integral), (int(flag.sum()), int(integral.sum()))
    ok = ~flag
    assert np.array_equal(offG[ok], ((-N) // 10000)[ok])
    assert not flag[GATE, GATE] and offG[GATE, GATE] == 0                     # U = V = 0: exempt, exact
    assert 80 <= int(flag.sum()) <= 130                                       # ~1e-4 of 1,050,625 pairs


def test_integer_offsets_reproduce_the_reference_formula_after_clamping():
    u = np.arange(-GATE, GATE + 1, dtype=np.int64)
    U, V = np.meshgrid(u, u, indexing="ij")
    offR, offG, offB, flag = kernel_offsets(U, V)
    Ud, Vd = U.astype(np.float64), V.astype(np.float64)
    pr, pg1, pg2, pb = Vd * 1.4020, Ud * 0.3441, Vd * 0.7139, Ud * 1.7718    # the reference's products (ref :567-578)

    def revise(v):                                                            # ref :672-676
        return np.where(v < 0.0, 0, np.where(v > 255.0, 255, np.trunc(np.clip(v, -1.0, 256.0)))).astype(np.int64)

    ys = list(range(-40, 300, 7)) + [-1024, -300, -129, -1, 0, 1, 127, 128, 254, 255, 256, 257, 511, 700, 1279]
    ok = ~flag
    for y in ys:
        yd = float(y)
        assert np.array_equal(revise(yd + pr), np.clip(y + offR, 0, 255)), y
        assert np.array_equal(revise(yd + pb), np.clip(y + offB, 0, 255)), y
        assert np.array_equal(revise(yd - pg1 - pg2)[ok], np.clip(y + offG, 0, 255)[ok]), y


def test_flagged_pairs_really_need_the_reference_sequence():
    """at least one flagged pair where floor(true value) and the reference's truncated double disagree for some Y: the flag is not
    a formality"""
    u = np.arange(-GATE, GATE + 1, dtype=np.int64)
    U, V = np.meshgrid(u, u, indexing="ij")
    _, offG, _, flag = kernel_offsets(U, V)
    fu, fv = U[flag], V[flag]
    N = 3441 * fu + 7139 * fv
    differs = 0
    for y in range(0, 256):
        ref = np.trunc(np.clip(float(y) - fu * 0.3441 - fv * 0.7139, -1.0, 256.0))
        differs += int(np.count_nonzero(ref != np.clip(y - N // 10000, -1, 256)))
    assert differs > 0
