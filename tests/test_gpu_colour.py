"""The decode kernel's colour conversion (step 5 of dequant_idct_kernel; ref decoder/jpezy_decoder.hpp:567-578, 672-676) on the GPU
against the oracle, aimed at the cases tests/test_colour_offsets.py reasons about: flat blocks place chosen integers (Y, Cb, Cr)
in front of the conversion --
  * every chroma pair inside the gate whose term 0.3441 U + 0.7139 V is a non-zero integer (the reference's own double
    sequence decides the last unit there: the kernel's waves with such a sample convert in doubles), and its lattice neighbours,
  * random pairs, gray (U = V = 0), pairs outside the gate (|U|, |V| up to 2000: also doubles),
each under sixteen luma values between -40 and 300.  Bit-exact in the default mode, within one in tolerance mode."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def J():
    import jpezy_amd
    jpezy_amd.load_library()
    return jpezy_amd


def _dc_for(sample):
    """a DC coefficient (quantiser 1) whose flat block decodes to `sample` under the reference's int(((S*S)*c)/4 + 128)"""
    t = sample - 128
    for c in (8 * t + 1, 8 * t - 1, 8 * t, 8 * t + 2, 8 * t - 2):
        if int((0.4999999999999999 * c) / 4 + 128) == sample:
            return c
    raise AssertionError(sample)


def test_flat_blocks_through_every_integral_chroma_term(J, oracle):
    GATE = 512
    u = np.arange(-GATE, GATE + 1)
    U, V = np.meshgrid(u, u, indexing="ij")
    N = 3441 * U + 7139 * V
    hit = (N % 10000 == 0) & (N != 0)
    pairs = [(int(a), int(b)) for a, b in zip(U[hit], V[hit])]
    assert len(pairs) > 80
    pairs += [(a + da, b + db) for a, b in pairs[::3] for da, db in ((1, 0), (0, 1), (-1, 0))]
    rng = np.random.default_rng(42)
    pairs += [(int(a), int(b)) for a, b in rng.integers(-140, 141, (200, 2))]
    pairs += [(0, 0), (0, 5), (7, 0), (-128, 127), (127, -128), (513, 3), (-700, 900), (2000, -2000), (40, 1999)]
    lumas = [-40, -3, -1, 0, 1, 2, 17, 64, 100, 127, 128, 200, 254, 255, 256, 300]
    mcus = [(p, lumas[4 * k:4 * k + 4]) for p in pairs for k in range(4)]
    mc = 64
    mr = (len(mcus) + mc - 1) // mc
    W, H = mc * 16, mr * 16
    co = np.zeros((mr, mc, 6, 64), np.int16)
    for i, ((pu, pv), ys) in enumerate(mcus):
        my, mx = divmod(i, mc)
        for b in range(4):
            co[my, mx, b, 0] = _dc_for(ys[b])
        co[my, mx, 4, 0] = _dc_for(pu + 128)
        co[my, mx, 5, 0] = _dc_for(pv + 128)
    qtab = type(J.api.annex_k_tables().qt)()
    info = oracle.make_info(W, H)
    for t in range(4):
        for k in range(64):
            qtab[t][k] = 1
            info.qt[t][k] = 1
    want = oracle.decode_planes(co, info, False)
    ctx = J.Context(0)
    try:
        got = ctx.dequant_idct(co, W, H, qt=qtab)
        for name, a, e in zip("rgb", got, want):
            bad = np.argwhere(a != e)
            assert bad.size == 0, (name, bad[:4], a[tuple(bad[0])], e[tuple(bad[0])])
        ctx.set_decode_tolerance(1)
        tol = ctx.dequant_idct(co, W, H, qt=qtab)
        assert max(int(np.abs(a.astype(np.int16) - e.astype(np.int16)).max()) for a, e in zip(tol, want)) <= 1
    finally:
        ctx.close()
    # the frame really holds the hard case: pixels where the exact value of g is an integer and the reference's doubles land below it
    below = 0
    for i, ((pu, pv), ys) in enumerate(mcus):
        n = 3441 * pu + 7139 * pv
        if n == 0 or n % 10000 or max(abs(pu), abs(pv)) > GATE:
            continue
        my, mx = divmod(i, mc)
        for b in range(4):
            exact = min(max(ys[b] - n // 10000, 0), 255)
            below += int(want[1].reshape(H, W)[my * 16 + (b >> 1) * 8, mx * 16 + (b & 1) * 8]) != exact
    assert below > 20
