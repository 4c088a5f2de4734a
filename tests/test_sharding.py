"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups run the frame sharding + coefficient gather
of jpezy_amd/sharding.py with the oracle standing in for the kernel (tests may use the oracle)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from jpezy_amd import sharding  # noqa: E402


def test_shard_range_partitions_exactly():
    for n in [0, 1, 5, 8, 4096, 4097]:
        for world in [1, 2, 3, 8]:
            spans = [sharding.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == sharding.max_shard(n, world)
    assert sharding.shard_range(4096, 8, 3) == (1536, 2048)          # BASELINE configs[3]: 512 frames per GPU
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, W, H, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        cpf = ((W + 15) // 16) * ((H + 15) // 16) * 6 * 64

        def encode_fn(lo, hi):
            co = [O.encode_coeffs(*O.synth_rgb(W, H, frame=f), W, H).reshape(-1) for f in range(lo, hi)]
            return torch.from_numpy(np.concatenate(co)) if co else torch.zeros(0, dtype=torch.int16)

        full = sharding.encode_batch_sharded(encode_fn, n_frames, cpf, gather=True)
        lo, hi, local = sharding.encode_batch_sharded(encode_fn, n_frames, cpf, gather=False)
        assert (lo, hi) == sharding.shard_range(n_frames, world, rank)
        assert torch.equal(full[lo:hi].reshape(-1), local)
        # max-over-ranks timing reduction as bench.py does it
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t[0]) == world
        np.save(Path(out_dir) / f"full_{rank}.npy", full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 5), (3, 4)])
def test_sharded_encode_and_gather_gloo(tmp_path, world, n_frames):
    W, H = 48, 32
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_frames, W, H, str(tmp_path)), nprocs=world, join=True)
    from oracle import oracle as O
    want = np.stack([O.encode_coeffs(*O.synth_rgb(W, H, frame=f), W, H).reshape(-1) for f in range(n_frames)])
    for r in range(world):
        got = np.load(tmp_path / f"full_{r}.npy")
        assert np.array_equal(got, want), f"rank {r} holds a wrong gathered batch"


def _worker_pipelined(rank, world, port, n_frames, chunk, W, H, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        cpf = ((W + 15) // 16) * ((H + 15) // 16) * 6 * 64
        calls = []

        def encode_chunk(lo, hi, dst):
            calls.append((lo, hi))
            my_lo, my_hi = sharding.shard_range(n_frames, world, rank)
            assert my_lo <= lo < hi <= my_hi and hi - lo <= chunk      # only this rank's frames, chunk by chunk
            for k, f in enumerate(range(lo, hi)):
                dst[k].copy_(torch.from_numpy(O.encode_coeffs(*O.synth_rgb(W, H, frame=f), W, H).reshape(-1)))

        full = sharding.gather_to_root_pipelined(encode_chunk, n_frames, cpf, chunk, "cpu", root=0, ring=2)
        if rank == 0:
            np.save(Path(out_dir) / "full.npy", full.numpy())
        else:
            assert full is None
        np.save(Path(out_dir) / f"calls_{rank}.npy", np.array(calls, dtype=np.int64).reshape(-1, 2))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames,chunk", [(2, 7, 2), (3, 8, 1), (3, 2, 4), (8, 19, 2)])
def test_pipelined_gather_to_root_gloo(tmp_path, world, n_frames, chunk):
    """chunked, overlapped gather-to-consumer: uneven shards, ragged last chunks, a rank with no frames,
    staging ring shorter than the number of chunks"""
    W, H = 32, 16
    port = _free_port()
    mp.spawn(_worker_pipelined, args=(world, port, n_frames, chunk, W, H, str(tmp_path)), nprocs=world, join=True)
    from oracle import oracle as O
    want = np.stack([O.encode_coeffs(*O.synth_rgb(W, H, frame=f), W, H).reshape(-1) for f in range(n_frames)])
    assert np.array_equal(np.load(tmp_path / "full.npy"), want)
    seen = []
    for r in range(world):
        calls = np.load(tmp_path / f"calls_{r}.npy")
        lo, hi = sharding.shard_range(n_frames, world, r)
        assert [tuple(c) for c in calls] == sharding.chunk_spans(lo, hi, chunk)
        seen += [f for a, b in calls for f in range(a, b)]
    assert sorted(seen) == list(range(n_frames))                      # every frame encoded exactly once


def test_pipelined_single_rank_no_process_group():
    cpf, n = 8, 5
    def enc(lo, hi, dst):
        dst.copy_(torch.arange(lo * cpf, hi * cpf, dtype=torch.int16).reshape(hi - lo, cpf))
    out = sharding.gather_to_root_pipelined(enc, n, cpf, 2, "cpu")
    assert torch.equal(out.reshape(-1), torch.arange(n * cpf, dtype=torch.int16))


def _worker_jpg(rank, world, port, n_frames, chunk, use_meta, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        meta = dist.new_group(backend="gloo") if use_meta else None
        stride = 4096
        stage = [torch.zeros((chunk, stride), dtype=torch.uint8) for _ in range(2)]
        calls = []

        def size_of(f):                     # frames of different sizes: variable-length files
            return 16 + 16 * (f % 3), 16 + 16 * (f % 2)

        def encode_chunk(lo, hi, slot):
            calls.append((lo, hi, slot))
            my_lo, my_hi = sharding.shard_range(n_frames, world, rank)
            assert my_lo <= lo < hi <= my_hi and hi - lo <= chunk and 0 <= slot < 2
            buf, sizes = stage[slot], torch.zeros(hi - lo, dtype=torch.int64)
            for k, f in enumerate(range(lo, hi)):
                W, H = size_of(f)
                jpg = O.encode_jpeg(*O.synth_rgb(W, H, frame=f), W, H, False)
                buf[k, : len(jpg)] = torch.frombuffer(bytearray(jpg), dtype=torch.uint8)
                sizes[k] = len(jpg)
            return buf[: hi - lo], sizes

        res = sharding.gather_jpg_to_root_pipelined(encode_chunk, n_frames, chunk, "cpu", root=0, meta_group=meta, ring=2)
        if rank == 0:
            np.save(Path(out_dir) / "sizes.npy", res.sizes.numpy())
            for f in range(n_frames):
                np.save(Path(out_dir) / f"jpg_{f}.npy", res.frame(f).numpy())
            assert res.total_bytes() == int(res.sizes.sum()) and [c[0] for c in res.chunks] == sorted(c[0] for c in res.chunks)
        else:
            assert res is None
        np.save(Path(out_dir) / f"calls_{rank}.npy", np.array(calls, dtype=np.int64).reshape(-1, 3))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames,chunk,use_meta", [(2, 7, 2, True), (3, 8, 1, False), (3, 2, 4, True), (2, 5, 3, False),
                                                           (8, 21, 2, True), (8, 11, 1, False)])
def test_pipelined_jpg_gather_to_root_gloo(tmp_path, world, n_frames, chunk, use_meta):
    """every rank codes its frames END TO END (the oracle stands in for FDCT + Huffman stage) and only the .jpg files --
    variable length -- travel to the consumer: uneven shards, ragged last chunks, a rank with no frames, a staging ring
    shorter than the number of chunks, sizes through a host-side gloo group or through the data group"""
    port = _free_port()
    mp.spawn(_worker_jpg, args=(world, port, n_frames, chunk, use_meta, str(tmp_path)), nprocs=world, join=True)
    from oracle import oracle as O
    sizes = np.load(tmp_path / "sizes.npy")
    for f in range(n_frames):
        W, H = 16 + 16 * (f % 3), 16 + 16 * (f % 2)
        want = O.encode_jpeg(*O.synth_rgb(W, H, frame=f), W, H, False)
        assert int(sizes[f]) == len(want)
        assert np.load(tmp_path / f"jpg_{f}.npy").tobytes() == want, f
    seen = []
    for r in range(world):
        calls = np.load(tmp_path / f"calls_{r}.npy")
        lo, hi = sharding.shard_range(n_frames, world, r)
        assert [(int(a), int(b)) for a, b, _ in calls] == sharding.chunk_spans(lo, hi, chunk)
        assert [int(s) for _, _, s in calls] == [c % 2 for c in range(len(calls))]
        seen += [f for a, b, _ in calls for f in range(a, b)]
    assert sorted(seen) == list(range(n_frames))


def test_pipelined_jpg_single_rank_no_process_group():
    def enc(lo, hi, slot):
        buf = torch.zeros((hi - lo, 16), dtype=torch.uint8)
        sizes = torch.tensor([1 + (f % 5) for f in range(lo, hi)], dtype=torch.int64)
        for k, f in enumerate(range(lo, hi)):
            buf[k, : int(sizes[k])] = f + 1
        return buf, sizes
    res = sharding.gather_jpg_to_root_pipelined(enc, 7, 3, "cpu")
    assert res.sizes.tolist() == [1 + (f % 5) for f in range(7)]
    for f in range(7):
        assert res.frame(f).tolist() == [f + 1] * (1 + f % 5)

    def bad(lo, hi, slot):
        return torch.zeros((hi - lo, 4), dtype=torch.uint8), torch.full((hi - lo,), -6, dtype=torch.int64)
    with pytest.raises(RuntimeError):
        sharding.gather_jpg_to_root_pipelined(bad, 2, 2, "cpu")


def _worker_banded(rank, world, port, W, H, gray, band_rows, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        r, g, b = O.synth_rgb(W, H, frame=3)
        mcu_cols, mcu_rows = (W + 15) // 16, (H + 15) // 16
        bands = []

        def encode_band(lo, hi, dst):
            my_lo, my_hi = sharding.shard_range(mcu_rows, world, rank)
            assert my_lo <= lo < hi <= my_hi
            y0, n = sharding.band_pixel_rows(lo, hi, H)
            bands.append((lo, hi, y0, n))
            # the band as a frame of its own: W x n pixels cut out of the planes
            sl = slice(y0 * W, (y0 + n) * W)
            co = O.encode_coeffs(r[sl], g[sl], b[sl], W, n, gray=gray)
            dst.copy_(torch.from_numpy(np.ascontiguousarray(co).reshape(hi - lo, -1)))

        full = sharding.encode_frame_banded(encode_band, W, H, gray=gray, band_rows=band_rows, device="cpu")
        if rank == 0:
            np.save(Path(out_dir) / "frame.npy", full.numpy())
        np.save(Path(out_dir) / f"bands_{rank}.npy", np.array(bands, dtype=np.int64).reshape(-1, 4))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,gray,band_rows", [(2, 48, 100, False, 2), (3, 33, 50, True, 1), (8, 64, 70, False, 1)])
def test_one_frame_split_by_mcu_row_bands_gloo(tmp_path, world, W, H, gray, band_rows):
    """a single frame over N ranks (configs[1] / [4] at N > 1): MCU-row bands transformed as frames of their own and gathered
    on rank 0 equal the one-rank frame bit for bit -- also the last, shorter band (bottom-edge replication), ranks with no
    band (more ranks than MCU rows: world 8, 5 MCU rows) and gray mode; the .jpg the root then writes is the one-rank file"""
    port = _free_port()
    mp.spawn(_worker_banded, args=(world, port, W, H, gray, band_rows, str(tmp_path)), nprocs=world, join=True)
    from oracle import oracle as O
    r, g, b = O.synth_rgb(W, H, frame=3)
    want = O.encode_coeffs(r, g, b, W, H, gray=gray)
    got = np.load(tmp_path / "frame.npy")
    assert np.array_equal(got.reshape(-1), np.ascontiguousarray(want).reshape(-1))
    assert O.write_jpeg(got.reshape(want.shape), W, H, gray) == O.encode_jpeg(r, g, b, W, H, gray)
    rows = []
    for k in range(world):
        for lo, hi, y0, n in np.load(tmp_path / f"bands_{k}.npy"):
            assert y0 == 16 * lo and n == min(16 * hi, H) - y0 and n > 0
            rows += list(range(lo, hi))
    assert sorted(rows) == list(range((H + 15) // 16))
