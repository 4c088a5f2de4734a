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
