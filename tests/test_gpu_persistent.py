"""Encode variant 2 on the GPU: variant 1's arithmetic in persistent workgroups with LDS-DMA loader waves
(jpezy_kernels_f32_ps.hip: fdct_quant_f32_ps_kernel, and variant 3, fdct_quant_f32_ps2_kernel: 16 compute waves per CU that
prefetch their next quad into registers).  Same bar as every encode test: the coefficients are the oracle's, bit
for bit, at every force_exact level.  What is specific to the persistent form is its hand-off machinery, so the cases
are chosen by the SHAPE OF THE WORK rather than by pixel content:

  * fewer groups than resident workgroups (one group per workgroup, workgroups with a single round),
  * many more groups than workgroups (every ring slot refilled many times, both loader waves busy, 4096 x 4096),
  * groups whose last quads or pieces are clamped (976 = 61 MCUs, 208 = 13 MCUs wide), a bottom band shorter than an MCU row,
  * several frames in one launch (the group index runs over frames; plane strides that are not W x H),
  * sizes the persistent form does not take (rows of quads that do not divide by four, unaligned planes): the launcher
    must hand them to variant 1's kernel,
  * every rare path inside the loop (force_exact 1/2/3; structured inputs full of guard-band hits): a wave that resolves
    hits must not disturb the waves it shares a ring with.

These kernels are the LABORATORY (round 5 measured them not faster than variant 1): the shipped library does not contain them and
jpezy_ctx_set_variant(ctx, 2) answers JPEZY_E_UNSUPPORTED -- this module then checks exactly that and skips the rest.  Against a
laboratory build (`python -m jpezy_amd._build --lab`, run with JPEZY_LIB=jpezy_amd/libjpezy_hip_lab.so) every case runs.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def J():
    import jpezy_amd
    jpezy_amd.load_library()
    return jpezy_amd


@pytest.fixture(scope="module", params=[2, 3], ids=["loader-waves", "register-prefetch"])
def ctx(J, request):
    c = J.Context(0)
    try:
        c.set_variant(request.param)
    except J.JpezyError as e:
        c.close()
        assert "laboratory" in str(e)
        pytest.skip("the shipped library holds no persistent encode kernels (laboratory builds only)")
    yield c
    c.close()


def test_the_shipped_library_refuses_the_laboratory_variants_loudly(J):
    """set_variant(2 / 3) either works (laboratory build) or fails with JPEZY_E_UNSUPPORTED and leaves the context on its kernel;
    never a silent switch, and no environment variable selects a kernel in the shipped build"""
    import os
    import subprocess
    import sys
    from jpezy_amd import api
    lib = api.load_library()
    c = J.Context(0)
    try:
        rcs = [lib.jpezy_ctx_set_variant(c._h, v) for v in (2, 3)]
        assert rcs in ([0, 0], [-4, -4])
        assert lib.jpezy_ctx_set_variant(c._h, 4) == -1 and lib.jpezy_ctx_set_variant(c._h, 1) == 0
    finally:
        c.close()
    if rcs == [-4, -4]:
        have = subprocess.run(["strings", str(J.library_path())], capture_output=True, text=True).stdout
        assert "fdct_quant_f32_ps" not in have and "JPEZY_ENC_VARIANT" not in have


@pytest.fixture(scope="module")
def ctx1(J):
    c = J.Context(0)
    c.set_variant(1)
    yield c
    c.close()


# (W, H): W % 16 == 0 and ceil(W / 64) % 4 == 0 take the persistent kernel
PERSISTENT_SIZES = [(256, 16), (256, 48), (1024, 16), (976, 33), (208, 40), (2048, 7), (768, 64), (4096, 64), (1024, 1024),
                    (7680, 32)]
OTHER_SIZES = [(64, 64), (1920, 24), (100, 100), (720, 486), (15, 17)]


@pytest.mark.parametrize("size", PERSISTENT_SIZES + OTHER_SIZES)
@pytest.mark.parametrize("gray", [False, True])
def test_encode_matches_oracle(J, ctx, oracle, size, gray):
    W, H = size
    r, g, b = oracle.synth_rgb(W, H, frame=W * 1000 + H + 5)
    want = oracle.encode_coeffs(r, g, b, W, H, gray)
    got = ctx.fdct_quant(r, g, b, W, H, gray=gray)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("force", [1, 2, 3])
def test_every_exactness_level_inside_the_loop(J, ctx, oracle, force):
    ctx.set_force_exact(force)
    try:
        for (W, H) in ((256, 32), (976, 33), (1024, 48)):
            r, g, b = oracle.synth_rgb(W, H, frame=3 * W + H)
            for gray in (False, True):
                want = oracle.encode_coeffs(r, g, b, W, H, gray)
                ctx.fallback_count()
                got = ctx.fdct_quant(r, g, b, W, H, gray=gray)
                assert np.array_equal(got, want), (W, H, gray, force)
                n = ctx.fallback_count()
                assert n < (1 << 40), "a wave of the persistent kernel gave up waiting (abort flag)"
                if force in (1, 2):
                    assert n == want.size
    finally:
        ctx.set_force_exact(0)


def test_golden_fixtures(J, ctx, golden_dir):
    for f in sorted(golden_dir.glob("*.npz")):
        z = np.load(f)
        W, H = int(z["W"]), int(z["H"])
        assert np.array_equal(ctx.fdct_quant(z["r"], z["g"], z["b"], W, H, gray=False), z["coeffs"]), f.stem
        assert np.array_equal(ctx.fdct_quant(z["r"], z["g"], z["b"], W, H, gray=True), z["coeffs_gray"]), f.stem


def test_structured_inputs_full_of_guard_band_hits(J, ctx, oracle):
    W, H = 256, 64
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:H, 0:W]
    imgs = [np.where((xx // 4 + yy // 4) % 2 == 0, 200, 40), np.where(xx % 8 < 4, 255, 0), ((xx // 16) * 16 + (yy // 16)) % 256,
            np.where((xx + yy) % 2 == 0, 255, 0), rng.integers(0, 2, (H, W)) * 255, rng.integers(0, 4, (H, W)) * 64 + 31,
            np.full((H, W), 128), np.zeros((H, W))]
    ctx.fallback_count()
    for k, im in enumerate(imgs):
        p = im.astype(np.uint8).reshape(-1)
        r, g, b = p, np.roll(p, k), p[::-1].copy()
        for gray in (False, True):
            want = oracle.encode_coeffs(r, g, b, W, H, gray)
            got = ctx.fdct_quant(r, g, b, W, H, gray=gray)
            assert np.array_equal(got, want), f"image {k} gray={gray}"
    n = ctx.fallback_count()
    assert 0 < n < (1 << 40)


def test_frames_in_one_launch_and_plane_strides(J, ctx, oracle):
    """The group index of the persistent kernel runs over all frames of the launch."""
    import torch
    W, H, F = 512, 80, 7
    frames = [oracle.synth_rgb(W, H, frame=140 + f) for f in range(F)]
    want = np.stack([oracle.encode_coeffs(*fr, W, H) for fr in frames])
    r, g, b = (np.concatenate([fr[k] for fr in frames]) for k in range(3))
    assert np.array_equal(ctx.fdct_quant(r, g, b, W, H, n_frames=F), want)

    dev = torch.device("cuda", 0)
    for stride, base_off in ((W * H + 48, 0), (W * H + 16 * 5, 16), (W * H + 40, 0), (W * H + 32, 8)):
        # strides / bases that are multiples of 16 stay on the persistent kernel, the others go to variant 1's unaligned form
        d = []
        for k in range(3):
            t = torch.zeros(F * stride + 64, dtype=torch.uint8, device=dev)
            t[base_off: base_off + F * stride].view(F, stride)[:, : W * H] = torch.from_numpy(np.stack([fr[k] for fr in frames])).to(dev)
            d.append(t[base_off:])
        dco = torch.full((F * J.coeff_count(W, H),), 77, dtype=torch.int16, device=dev)
        ctx.fdct_quant_dev(d[0], d[1], d[2], W, H, dco, n_frames=F, plane_stride=stride)
        torch.cuda.synchronize()
        assert np.array_equal(dco.cpu().numpy().reshape(want.shape), want), (stride, base_off)


def test_many_more_groups_than_workgroups_4096(J, ctx, ctx1, oracle):
    """BASELINE configs[1]: 4,096 groups over 512 resident workgroups; whole frame against the oracle (banded over host
    threads) and against variant 1's kernel; twice in a row on the same context (control words start from zero again)."""
    from test_gpu_parity import _threaded_oracle
    W = H = 4096
    r, g, b = oracle.synth_rgb(W, H, frame=4097)
    mc, mr = J.mcu_grid(W, H)
    want = np.zeros((mr, mc, 6, 64), np.int16)
    lib = oracle.lib()

    def band(rows):
        lib.jo_encode_coeffs_rows(oracle._u8(r), oracle._u8(g), oracle._u8(b), W, H, 0, rows[0], rows[1], oracle._i16(want))
    _threaded_oracle(band, mr)
    ctx.fallback_count()
    got = ctx.fdct_quant(r, g, b, W, H)
    assert np.array_equal(got, want)
    n2 = ctx.fallback_count()
    ctx1.fallback_count()
    assert np.array_equal(ctx1.fdct_quant(r, g, b, W, H), want)
    assert n2 == ctx1.fallback_count()                 # the same coefficients took the exact path in both forms
    assert np.array_equal(ctx.fdct_quant(r, g, b, W, H), want)


def test_gray_8k_and_device_resident_replays(J, ctx, ctx1, oracle):
    """BASELINE configs[4] geometry (7680 wide: 30 groups per row) and the benchmark's way of calling: device-resident
    planes, many launches back to back on one stream."""
    import torch
    W, H = 7680, 4320
    r, g, b = oracle.synth_rgb(W, 540, frame=9)
    r, g, b = (np.tile(p, 8) for p in (r, g, b))
    want = ctx1.fdct_quant(r, g, b, W, H, gray=True)
    assert np.array_equal(ctx.fdct_quant(r, g, b, W, H, gray=True), want)
    W = H = 2048
    dev = torch.device("cuda", 0)
    frames = [oracle.synth_rgb(W, H, frame=300 + f) for f in range(3)]
    d = [[torch.from_numpy(fr[k]).to(dev) for k in range(3)] for fr in frames]
    outs = [torch.empty(J.coeff_count(W, H), dtype=torch.int16, device=dev) for _ in range(3)]
    for rep in range(20):
        for f in range(3):
            ctx.fdct_quant_dev(d[f][0], d[f][1], d[f][2], W, H, outs[f])
    torch.cuda.synchronize()
    for f in range(3):
        assert np.array_equal(outs[f].cpu().numpy(), ctx1.fdct_quant(*frames[f], W, H).reshape(-1))
