"""Streaming host-buffer entry points (jpezy_hostpipe.h): jpezy_fdct_quant / jpezy_dequant_idct / jpezy_encode_jpeg cut a call
into chunks that flow through a ring of four pinned staging slots.  Forced to tiny chunks here so that a call has many more
chunks than the ring has slots -- bands of one frame and groups of frames, ragged sizes, colour and gray -- and compared with
the oracle (ref encoder/jpezy_encoder.hpp:58-67, decoder/jpezy_decoder.hpp:504-578)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def J():
    import jpezy_amd
    jpezy_amd.load_library()
    return jpezy_amd


@pytest.mark.parametrize("chunk", [4096, 60000, 4 << 20])
def test_bands_of_one_frame(J, oracle, chunk):
    ctx = J.Context(0)
    try:
        ctx.set_host_chunk_bytes(chunk)
        for (W, H) in ((640, 480), (333, 517), (16, 700)):
            r, g, b = oracle.synth_rgb(W, H, frame=W + H)
            for gray in (False, True):
                want = oracle.encode_coeffs(r, g, b, W, H, gray)
                got = ctx.fdct_quant(r, g, b, W, H, gray=gray)
                assert np.array_equal(got, want), (W, H, gray, chunk)
                assert ctx.encode_jpeg(r, g, b, W, H, gray=gray) == oracle.encode_jpeg(r, g, b, W, H, gray)
            co = oracle.encode_coeffs(r, g, b, W, H)
            for gray in (False, True):
                for a, e in zip(ctx.dequant_idct(co, W, H, gray=gray), oracle.decode_planes(co, oracle.make_info(W, H), gray)):
                    assert np.array_equal(a, e), (W, H, gray, chunk)
    finally:
        ctx.close()


@pytest.mark.parametrize("chunk,F", [(4096, 23), (100000, 23), (4 << 20, 9)])
def test_more_frames_than_ring_slots(J, oracle, chunk, F):
    """n_frames far larger than the ring (4 slots): groups of frames per chunk, last group ragged"""
    W, H = 96, 80
    ctx = J.Context(0)
    try:
        ctx.set_host_chunk_bytes(chunk)
        frames = [oracle.synth_rgb(W, H, frame=700 + f) for f in range(F)]
        want = np.stack([oracle.encode_coeffs(*fr, W, H) for fr in frames])
        r, g, b = (np.concatenate([fr[k] for fr in frames]) for k in range(3))
        got = ctx.fdct_quant(r, g, b, W, H, n_frames=F)
        assert np.array_equal(got, want)
        dr, dg, db = ctx.dequant_idct(want, W, H, n_frames=F)
        info = oracle.make_info(W, H)
        for f in (0, 1, F // 2, F - 1):
            ref = oracle.decode_planes(want[f], info)
            for a, e in zip((dr, dg, db), ref):
                assert np.array_equal(a.reshape(F, -1)[f], e), f
    finally:
        ctx.close()


def test_fresh_buffers_every_call_and_changing_sizes(J, oracle):
    """the ring is regrown when a later call needs larger slots; results do not depend on what ran before"""
    ctx = J.Context(0)
    try:
        for (W, H) in ((64, 64), (1920, 1080), (100, 60), (2048, 2048)):
            r, g, b = (np.array(p) for p in oracle.synth_rgb(W, min(H, 270), frame=W))
            reps = -(-H // min(H, 270))
            r, g, b = (np.ascontiguousarray(np.tile(p, reps)[: W * H]) for p in (r, g, b))
            got = ctx.fdct_quant(r, g, b, W, H)
            mc, mr = J.mcu_grid(W, H)
            want = np.zeros((mr, mc, 6, 64), np.int16)
            lib = oracle.lib()
            lib.jo_encode_coeffs_rows(oracle._u8(r), oracle._u8(g), oracle._u8(b), W, H, 0, 0, mr, oracle._i16(want))
            assert np.array_equal(got, want), (W, H)
    finally:
        ctx.close()
