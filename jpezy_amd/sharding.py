"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).

Frames of a batch are independent (the only cross-MCU state of the reference, pre_DC and the bit cursor, lives
in the per-frame serial Huffman tail), so the hot path shards with NO data-path collective: rank k encodes the
contiguous frame range shard_range(n_frames, world, k).  BASELINE.json's north_star additionally asks for the
int16 coefficient buffers to be gathered over xGMI; gather_coefficients() does that with one all_gather of
equal, padded chunks (8 ranks x 3.2 GB for the 4096-frame 1080p batch -- bounded by the 7 x ~153 GB/s links of
the receiving GPU, so it is reported separately from the kernel throughput).
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
