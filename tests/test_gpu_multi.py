"""jpezy_encode_batch_multi on the GPU box (one GPU): with one shard the results must be those of the single-frame entry points, frame
by frame; the multi-shard path -- worker threads, shard arithmetic, two contexts and streams per shard, chunk pipelining, the gather
into the root device's memory with hipMemcpyPeerAsync or into host memory -- runs with several shards on the same device index.
Traffic between two different GPUs has never run (no such box is offered): DESIGN.md section 8 says so."""
import os
import subprocess

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


def _frames(oracle, W, H, F, seed):
    frames = [oracle.synth_rgb(W, H, frame=seed + f) for f in range(F)]
    return frames, [np.concatenate([fr[k] for fr in frames]) for k in range(3)]


@pytest.mark.parametrize("gray", [False, True])
def test_one_shard_equals_the_single_frame_entry_points(J, ctx, oracle, gray):
    W, H, F = 208, 120, 5
    frames, (r, g, b) = _frames(oracle, W, H, F, 900)
    co, jpg = J.encode_batch_multi([0], r, g, b, W, H, F, gray=gray, chunk_frames=2, want_coeffs=True)
    for f, fr in enumerate(frames):
        assert np.array_equal(co[f], ctx.fdct_quant(*fr, W, H, gray=gray).reshape(-1)), f
        assert jpg[f] == ctx.encode_jpeg(*fr, W, H, gray=gray), f
        assert jpg[f] == oracle.write_jpeg(oracle.encode_coeffs(*fr, W, H, gray), W, H, gray)


@pytest.mark.parametrize("on_root", [False, True])
@pytest.mark.parametrize("shards,F,chunk", [(2, 7, 2), (3, 7, 1), (4, 3, 4), (3, 20, 3), (8, 8, 0)])
def test_several_shards_on_one_device(J, ctx, oracle, shards, F, chunk, on_root):
    """more shards than frames (empty shards), chunk sizes that do not divide a shard, both placements of the results"""
    W, H = 256, 80
    frames, (r, g, b) = _frames(oracle, W, H, F, 1000 + shards)
    co, jpg = J.encode_batch_multi([0] * shards, r, g, b, W, H, F, chunk_frames=chunk, want_coeffs=True, on_root_device=on_root)
    want = np.stack([oracle.encode_coeffs(*fr, W, H) for fr in frames])
    assert np.array_equal(co.reshape(want.shape), want)
    for f in range(F):
        assert jpg[f] == oracle.write_jpeg(want[f], W, H, False), f


def test_only_one_of_the_two_outputs_and_a_stride_too_small(J, oracle):
    W, H, F = 128, 64, 4
    frames, (r, g, b) = _frames(oracle, W, H, F, 77)
    co, jpg = J.encode_batch_multi([0, 0], r, g, b, W, H, F, want_coeffs=True, want_jpg=False)
    assert jpg is None and np.array_equal(co[3], oracle.encode_coeffs(*frames[3], W, H).reshape(-1))
    co, jpg = J.encode_batch_multi([0, 0], r, g, b, W, H, F, want_coeffs=False, want_jpg=True)
    assert co is None and jpg[0] == oracle.write_jpeg(oracle.encode_coeffs(*frames[0], W, H), W, H, False)
    for on_root in (False, True):
        co, jpg = J.encode_batch_multi([0, 0], r, g, b, W, H, F, jpg_stride=1024, on_root_device=on_root)   # random pixels need ~6 KB per frame
        assert all(j == -6 for j in jpg)                                                                     # JPEZY_E_NOSPACE per frame, nothing else written


def test_bigger_frames_many_chunks(J, ctx, oracle):
    """BASELINE configs[3] geometry in small: 1920x1080 frames, 3 shards, chunks of 2"""
    W, H, F = 1920, 1080, 9
    frames, (r, g, b) = _frames(oracle, W, H, F, 5)
    co, jpg = J.encode_batch_multi([0, 0, 0], r, g, b, W, H, F, chunk_frames=2, want_coeffs=True, on_root_device=True)
    for f in (0, 4, 8):
        assert np.array_equal(co[f], ctx.fdct_quant(*frames[f], W, H).reshape(-1))
        assert jpg[f] == ctx.encode_jpeg(*frames[f], W, H)


def test_handle_reuse_with_changing_frame_counts(J, ctx, oracle):
    """jpezy_multi_create once, jpezy_multi_encode many times: frame counts that grow, shrink, leave lanes empty, do not divide by the
    chunk; outputs switch between host and root-device memory, coefficients and files; every call equal to the single-frame entries"""
    import torch
    W, H = 208, 120
    frames, (r, g, b) = _frames(oracle, W, H, 23, 4000)
    want_co = [ctx.fdct_quant(*fr, W, H).reshape(-1) for fr in frames]
    want_jpg = [ctx.encode_jpeg(*fr, W, H) for fr in frames]
    torch.cuda.set_device(0)
    with J.MultiEncoder([0, 0, 0], W, H, chunk_frames=2) as M:
        assert M.chunk_frames == 2
        for n, on_root, co_too in [(7, False, True), (23, False, False), (1, False, True), (2, True, True), (23, True, True), (5, False, True),
                                   (3, True, False)]:
            px = [p[: n * W * H] for p in (r, g, b)]
            co, jpg = M.encode(*px, n, want_coeffs=co_too, on_root_device=on_root)
            assert [j for j in jpg] == want_jpg[:n], (n, on_root)
            if co_too:
                assert np.array_equal(co, np.stack(want_co[:n])), (n, on_root)
            st = M.stats()
            assert len(st) == 3 and sum(s["frames"] for s in st) == n
            assert sum(s["bytes_up"] for s in st) == 3 * W * H * n
            assert all(s["staged"] == 1 for s in st if s["frames"])                     # numpy memory is pageable: staged through the ring
            assert all(s["kernel_ms"] > 0 and s["wall_ms"] >= s["kernel_ms"] * 0.5 for s in st if s["frames"])
        assert torch.cuda.current_device() == 0
    # a gray handle, default chunk size
    with J.MultiEncoder([0, 0], W, H, gray=True) as M:
        assert M.chunk_frames >= 1
        for n in (3, 9):
            co, jpg = M.encode(r[: n * W * H], g[: n * W * H], b[: n * W * H], n, want_coeffs=True)
            for f in range(n):
                assert np.array_equal(co[f], ctx.fdct_quant(*frames[f], W, H, gray=True).reshape(-1))
                assert jpg[f] == ctx.encode_jpeg(*frames[f], W, H, gray=True)


def test_planes_the_caller_has_pinned_are_uploaded_as_they_are(J, ctx, oracle):
    """pinned host memory (torch pin_memory = hipHostMalloc) goes to the DMA engines without the staging copy; same results"""
    import torch
    W, H, F = 256, 80, 11
    frames, planes = _frames(oracle, W, H, F, 5000)
    pinned = [torch.from_numpy(p.copy()).pin_memory() for p in planes]
    with J.MultiEncoder([0, 0], W, H, chunk_frames=3) as M:
        co, jpg = M.encode(*pinned, F, want_coeffs=True, raw=False)
        assert all(s["staged"] == 0 for s in M.stats() if s["frames"])
        co2, jpg2 = M.encode(*planes, F, want_coeffs=True)                               # the same handle, pageable planes
        assert all(s["staged"] == 1 for s in M.stats() if s["frames"])
    assert np.array_equal(co, co2) and jpg == jpg2
    for f in (0, 5, 10):
        assert jpg[f] == ctx.encode_jpeg(*frames[f], W, H)


def test_a_failing_call_leaves_the_handle_usable(J, ctx, oracle):
    W, H, F = 128, 64, 6
    frames, (r, g, b) = _frames(oracle, W, H, F, 6000)
    with J.MultiEncoder([0, 0], W, H, chunk_frames=2) as M:
        co, jpg = M.encode(r, g, b, F, jpg_stride=1024)                                  # every frame refused: JPEZY_E_NOSPACE each
        assert all(j == -6 for j in jpg)
        co, jpg = M.encode(r, g, b, F)
        assert jpg == [ctx.encode_jpeg(*fr, W, H) for fr in frames]
        with pytest.raises(J.JpezyError):
            M.encode(r[:-1], g, b, F)


def test_512_frames_1080p_through_the_handle(J, ctx, oracle):
    """a shard of BASELINE configs[3] as ONE call of the native handle (what one GPU of an 8-GPU run is handed): 512 frames 1920x1080 from
    pageable host planes, two lanes on the one device, default chunking -- the rings wrap some sixty times; every file's size is checked,
    a sample of frames byte for byte against the single-frame entry and the oracle, and the lanes' statistics must add up"""
    W, H, F, distinct = 1920, 1080, 512, 8
    base = [oracle.synth_rgb(W, H, frame=9000 + k) for k in range(distinct)]
    planes = [np.ascontiguousarray(np.tile(np.stack([b[q] for b in base]), (F // distinct, 1))).reshape(-1) for q in range(3)]
    want = [ctx.encode_jpeg(*base[k], W, H) for k in range(distinct)]
    assert want[3] == oracle.write_jpeg(oracle.encode_coeffs(*base[3], W, H), W, H, False)
    with J.MultiEncoder([0, 0], W, H) as M:
        _, jpg = M.encode(*planes, F, jpg_stride=1 << 20)
        st = M.stats()
    assert len(jpg) == F and [len(j) for j in jpg] == [len(want[f % distinct]) for f in range(F)]
    for f in (0, 1, 7, 8, 255, 256, 257, 300, 510, 511):
        assert jpg[f] == want[f % distinct], f
    assert [s_["frames"] for s_ in st] == [256, 256] and sum(s_["bytes_up"] for s_ in st) == 3 * W * H * F
    assert sum(s_["bytes_down"] for s_ in st) == sum(len(j) for j in jpg)


def test_the_callers_current_device_is_put_back(J, oracle):
    import ctypes as C
    import torch
    from jpezy_amd import api
    hip = C.CDLL("libamdhip64.so")
    torch.cuda.set_device(0)
    W, H, F = 64, 48, 3
    _, (r, g, b) = _frames(oracle, W, H, F, 1)
    J.encode_batch_multi([0, 0], r, g, b, W, H, F)
    d = C.c_int(-1)
    assert hip.hipGetDevice(C.byref(d)) == 0 and d.value == 0


def test_cli_gpus_mode(J, oracle, tmp_path):
    """jpezy_encode --gpus N in out [in out ...]: runs of one size form a batch; the files equal those of the one-file CLI form"""
    from pathlib import Path
    exe = Path(J.library_path()).parent / "bin" / "jpezy_encode"
    assert exe.exists()
    sizes = [(64, 48), (64, 48), (80, 32), (64, 48)]
    args = []
    for i, (W, H) in enumerate(sizes):
        r, g, b = oracle.synth_rgb(W, H, frame=60 + i)
        ppm = tmp_path / f"in{i}.ppm"
        with open(ppm, "w") as f:
            f.write(f"P3\n{W} {H}\n255\n")
            f.write("".join(f"{int(x)} {int(y)} {int(z)}\n" for x, y, z in zip(r, g, b)))
        args += [str(ppm), str(tmp_path / f"out{i}.jpg")]
    p = subprocess.run([str(exe), "--gpus", "8", *args], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "Encoded 4 file(s) on 1 GPU(s)" in p.stdout
    for i, (W, H) in enumerate(sizes):
        r, g, b = oracle.synth_rgb(W, H, frame=60 + i)
        want = oracle.write_jpeg(oracle.encode_coeffs(r, g, b, W, H), W, H, False)
        assert (tmp_path / f"out{i}.jpg").read_bytes() == want, i
    p = subprocess.run([str(exe), "--gpus", "2", "--gray", args[0], str(tmp_path / "g.jpg")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0
    r, g, b = oracle.synth_rgb(64, 48, frame=60)
    assert (tmp_path / "g.jpg").read_bytes() == oracle.write_jpeg(oracle.encode_coeffs(r, g, b, 64, 48, True), 64, 48, True)
    assert subprocess.run([str(exe), "--gpus", "0", args[0], args[1]], capture_output=True).returncode == 1
    assert subprocess.run([str(exe), "--gpus", "2", args[0]], capture_output=True).returncode == 1
