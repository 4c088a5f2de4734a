"""GPU entropy coder (Huffman coding, bit packing and byte stuffing on the device) against the host
writer and the oracle: identical .jpg bytes for the golden fixtures, stress coefficient patterns (long zero runs, ZRL
chains, maximal categories, streams full of 0xFF bytes), gray mode, ragged sizes, batches and the full 4096x4096 frame."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def J():
    import jpezy_amd
    jpezy_amd.load_library()
    return jpezy_amd


@pytest.fixture(scope="module")
def ctx(J):
    c = J.Context(0)
    yield c
    c.close()


def _dev(co):
    import torch
    return torch.from_numpy(np.ascontiguousarray(co, dtype=np.int16)).cuda()


def _stress_coeffs(rng, nmcu, bpm=6):
    co = np.zeros((nmcu, bpm, 64), np.int16)
    for m in range(nmcu):
        for b in range(bpm):
            blk = co[m, b]
            kind = rng.integers(0, 8)
            if kind == 0:
                pass                                        # all zero: DC diff + EOB
            elif kind == 1:
                blk[0] = rng.integers(-1023, 1024)
                blk[63] = rng.integers(1, 1024)             # 62 zeros then a value: three ZRLs, no EOB
            elif kind == 2:
                blk[:] = rng.integers(-1023, 1024, 64)      # dense, maximal categories
            elif kind == 3:
                blk[0] = -1023
                blk[1:] = 1023                              # long runs of 1-bits: 0xFF bytes to stuff
            elif kind == 4:
                blk[rng.integers(1, 64, 5)] = rng.integers(-7, 8, 5)
            elif kind == 5:
                blk[17] = 1; blk[34] = -1; blk[51] = 2      # runs of exactly 16: ZRL + run 0
            elif kind == 6:
                blk[0] = rng.integers(-1023, 1024)
                blk[16] = -512                              # run 15 (no ZRL), size 10
            else:
                blk[:] = rng.integers(-3, 4, 64)
    return co


def test_fixtures_byte_identical(J, ctx, golden_dir):
    for path in sorted(golden_dir.glob("*.npz")):
        z = np.load(path)
        W, H = int(z["W"]), int(z["H"])
        assert ctx.write_jpeg_gpu(_dev(z["coeffs"]), W, H)[0] == z["jpg"].tobytes(), path.name
        assert ctx.write_jpeg_gpu(_dev(z["coeffs_gray"]), W, H, gray=True)[0] == z["jpg_gray"].tobytes(), path.name


@pytest.mark.parametrize("size", [(16, 16), (17, 17), (48, 32), (100, 60), (640, 480), (1920, 1080)])
def test_stress_coefficients_match_host_writer_and_oracle(J, ctx, oracle, size):
    W, H = size
    rng = np.random.default_rng(W * 131 + H)
    mc, mr = J.mcu_grid(W, H)
    co = _stress_coeffs(rng, mc * mr).reshape(mr, mc, 6, 64)
    want = J.write_jpeg(co, W, H, False)
    assert ctx.write_jpeg_gpu(_dev(co), W, H)[0] == want
    if W * H <= 100 * 60:
        assert want == oracle.write_jpeg(co, W, H, False)
    g = np.ascontiguousarray(co[:, :, :4])
    assert ctx.write_jpeg_gpu(_dev(g), W, H, gray=True)[0] == J.write_jpeg(g, W, H, True)
    assert ctx.write_jpeg_gpu(_dev(co), W, H, comment=b"")[0] == J.write_jpeg(co, W, H, False, comment=b"")


def test_byte_stuffing_heavy_stream(J, ctx):
    co = np.zeros((4, 4, 6, 64), np.int16)
    co[..., 0] = -1023
    co[..., 1:] = 1023
    jpg = ctx.write_jpeg_gpu(_dev(co), 64, 64)[0]
    assert jpg == J.write_jpeg(co, 64, 64)
    ent = jpg[644:-2]
    assert ent.count(b"\xff\x00") > 100


def test_out_of_table_coefficients_are_an_error(J, ctx):
    co = np.zeros((1, 1, 6, 64), np.int16)
    co[0, 0, 0, 5] = 1024
    with pytest.raises(J.JpezyError):
        ctx.write_jpeg_gpu(_dev(co), 16, 16)
    co[0, 0, 0, 5] = 0
    co[0, 0, 0, 0] = 2048
    with pytest.raises(J.JpezyError):
        ctx.write_jpeg_gpu(_dev(co), 16, 16)


def test_batch_and_end_to_end(J, ctx, oracle):
    W, H, n = 208, 120, 5
    frames = [oracle.synth_rgb(W, H, frame=f) for f in range(n)]
    cos = np.stack([ctx.fdct_quant(r, g, b, W, H) for (r, g, b) in frames])
    got = ctx.write_jpeg_gpu(_dev(cos), W, H, n_frames=n)
    for f in range(n):
        assert got[f] == J.write_jpeg(cos[f], W, H), f
        r, g, b = frames[f]
        assert ctx.encode_jpeg(r, g, b, W, H) == got[f]
        assert got[f] == oracle.encode_jpeg(r, g, b, W, H, False)
    r, g, b = frames[0]
    assert ctx.encode_jpeg(r, g, b, W, H, gray=True) == oracle.encode_jpeg(r, g, b, W, H, True)


def test_full_4096_frame(J, ctx, oracle):
    import torch
    W = H = 4096
    r, g, b = oracle.synth_rgb(W, H, frame=0)
    co = ctx.fdct_quant(r, g, b, W, H)
    want = J.write_jpeg(co, W, H)
    got = ctx.write_jpeg_gpu(_dev(co), W, H)[0]
    assert len(got) == len(want) and hashlib.sha256(got).digest() == hashlib.sha256(want).digest()
    torch.cuda.synchronize()
