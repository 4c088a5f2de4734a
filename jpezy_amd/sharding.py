"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).

Frames of a batch are independent (the only cross-MCU state of the reference, pre_DC and the bit cursor, lives
in the per-frame serial Huffman tail; the loop being sharded is the MCU loop of encoder::encode, ref
encoder/jpezy_encoder.hpp:55-67, run once per frame), so the hot path shards with NO data-path collective:
rank k encodes the contiguous frame range shard_range(n_frames, world, k).  BASELINE.json's north_star
additionally asks for the int16 coefficient buffers to be gathered over xGMI:

  gather_to_root_pipelined()  the production form: frames are encoded in chunks of `chunk_frames`; while chunk c+1
                              is being encoded on the compute stream, chunk c travels to the consumer rank (the
                              one that writes the files) as point-to-point sends on a side stream.  Only the
                              consumer holds the whole batch (7 x 3.2 GB into one GPU at 8 ranks is
                              bounded by that GPU's 7 xGMI links, so it is overlapped, not waited for).
  gather_coefficients()       one monolithic all_gather after all kernels (every rank ends up with the batch);
                              kept for callers that want replicas, and as the un-overlapped comparison.
"""
import torch
import torch.distributed as dist


def shard_range(n_units, world, rank):
    """Contiguous [lo, hi) of `n_units` for `rank`; sizes differ by at most one, earlier ranks get the extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_shard(n_units, world):
    return -(-n_units // world)


def chunk_spans(lo, hi, chunk_frames):
    """[lo, hi) cut into consecutive spans of at most chunk_frames frames."""
    if chunk_frames <= 0:
        raise ValueError("chunk_frames must be positive")
    return [(a, min(a + chunk_frames, hi)) for a in range(lo, hi, chunk_frames)]


def gather_coefficients(local, n_frames, coeffs_per_frame, group=None):
    """local: int16 tensor [local_frames * coeffs_per_frame] of this rank's shard (its device decides the backend
    path).  Returns the [n_frames, coeffs_per_frame] tensor of the whole batch on every rank."""
    world = dist.get_world_size(group)
    cap = max_shard(n_frames, world) * coeffs_per_frame
    padded = local
    if local.numel() != cap:
        padded = torch.zeros(cap, dtype=local.dtype, device=local.device)
        padded[: local.numel()] = local.reshape(-1)
    out = torch.empty(world * cap, dtype=local.dtype, device=local.device)
    # moved as bytes: the gloo backend of the CPU tests has no int16 collectives, RCCL does not care
    dist.all_gather_into_tensor(out.view(torch.uint8), padded.reshape(-1).view(torch.uint8), group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_frames, world, r)
        parts.append(out[r * cap: r * cap + (hi - lo) * coeffs_per_frame])
    return torch.cat(parts).reshape(n_frames, coeffs_per_frame)


def encode_batch_sharded(encode_fn, n_frames, coeffs_per_frame, gather=True, group=None):
    """encode_fn(lo, hi) -> int16 tensor with the coefficients of frames [lo, hi) (on the GPU path: a call of
    Context.fdct_quant_dev on this rank's device).  Returns (lo, hi, local) or the gathered batch."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n_frames, world, rank)
    local = encode_fn(lo, hi)
    if local.numel() != (hi - lo) * coeffs_per_frame:
        raise ValueError("encode_fn returned a buffer of the wrong size")
    if not gather or world == 1:
        return lo, hi, local
    return gather_coefficients(local, n_frames, coeffs_per_frame, group)


def gather_to_root_pipelined(encode_chunk, n_frames, coeffs_per_frame, chunk_frames, device, root=0, group=None,
                             out=None, ring=3, dtype=torch.int16):
    """Sharded encode with the coefficient gather overlapped chunk by chunk.

    encode_chunk(lo, hi, dst): enqueue (on the CURRENT stream of `device`, or compute synchronously on the CPU) the
        encode of global frames [lo, hi) -- all inside this rank's shard -- into dst, an int16 tensor
        [hi - lo, coeffs_per_frame].
    Returns the [n_frames, coeffs_per_frame] batch on `root`, None elsewhere.  `out` may be a preallocated result
    tensor on root.  Non-root ranks stage chunks in a ring of `ring` buffers; a buffer is reused only after its send
    has completed.  With one rank (or no process group) this degenerates to encoding straight into `out`.
    """
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    world = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    on_gpu = torch.device(device).type == "cuda"
    spans = [chunk_spans(*shard_range(n_frames, world, r), chunk_frames) for r in range(world)]
    rounds = max(len(s) for s in spans) if spans else 0

    if rank == root:
        if out is None:
            out = torch.empty((n_frames, coeffs_per_frame), dtype=dtype, device=device)
        elif tuple(out.shape) != (n_frames, coeffs_per_frame):
            raise ValueError("out has the wrong shape")
    if not multi:
        for lo, hi in spans[0]:
            encode_chunk(lo, hi, out[lo:hi])
        return out

    def as_bytes(t):            # gloo has no int16 point-to-point; bytes travel the same on RCCL
        return t.reshape(-1).view(torch.uint8)

    comm = torch.cuda.Stream(device) if on_gpu else None
    stage, pending = [], []     # non-root: staging ring and the work handles that still read each slot
    if rank != root:
        stage = [torch.empty((chunk_frames, coeffs_per_frame), dtype=dtype, device=device) for _ in range(ring)]
        pending = [None] * ring
    works = []
    for c in range(rounds):
        mine = spans[rank][c] if c < len(spans[rank]) else None
        ev = None
        if rank == root:
            if mine:
                encode_chunk(mine[0], mine[1], out[mine[0]:mine[1]])
            ops = []
            for r in range(world):
                if r != root and c < len(spans[r]):
                    lo, hi = spans[r][c]
                    ops.append(dist.P2POp(dist.irecv, as_bytes(out[lo:hi]), r, group))
        else:
            ops = []
            if mine:
                slot = c % ring
                if pending[slot] is not None:          # the send that last read this slot must be done
                    for w in pending[slot]:
                        w.wait()
                        works.remove(w)                # (waiting twice on a finished gloo send never returns)
                    pending[slot] = None
                buf = stage[slot][: mine[1] - mine[0]]
                encode_chunk(mine[0], mine[1], buf)
                if on_gpu:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(device))
                ops.append(dist.P2POp(dist.isend, as_bytes(buf), root, group))
        if ops:
            if on_gpu:
                # the transfer waits for this chunk's kernels only; the compute stream goes on with chunk c + 1
                if ev is not None:
                    comm.wait_event(ev)
                with torch.cuda.stream(comm):
                    w = dist.batch_isend_irecv(ops)
            else:
                w = dist.batch_isend_irecv(ops)
            works.extend(w)
            if rank != root and mine:
                pending[c % ring] = w
    for w in works:
        w.wait()
    if on_gpu:
        torch.cuda.current_stream(device).wait_stream(comm)
    return out if rank == root else None
