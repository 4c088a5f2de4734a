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
  gather_jpg_to_root_pipelined()  encoder::encode END TO END on every rank (ref encoder/jpezy_encoder.hpp:38-77: MCU loop AND
                              Huffman tail, both on the rank's GPU) and only the finished .jpg files travel to the consumer:
                              a tenth of the coefficient bytes on noise, far less on pictures -- the pipeline whose 8-GPU
                              speed-up is not capped by the consumer's seven xGMI links.
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
    if on_gpu:
        # `out` may be fresh from the caching allocator or still read by kernels queued on the compute stream (a previous
        # call's consumers): the incoming transfers must not overtake them.  (The result is stream-ordered on return -- the
        # compute stream waits for the side stream below -- with no host synchronisation.)
        comm.wait_stream(torch.cuda.current_stream(device))
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


def band_pixel_rows(mcu_lo, mcu_hi, H):
    """(first pixel row, number of pixel rows) of the band of MCU rows [mcu_lo, mcu_hi) of a frame H pixels high.  A band of
    whole MCU rows is a frame of its own for the kernels: only the last band is shorter than 16 rows per MCU row, and it
    replicates the bottom edge exactly as the whole frame would (ref encoder/jpezy_encoder.hpp:101)."""
    y0 = 16 * mcu_lo
    return y0, min(16 * mcu_hi, H) - y0


def encode_frame_banded(encode_band, W, H, gray=False, band_rows=8, device="cpu", root=0, group=None, out=None, ring=3):
    """ONE frame over all ranks (BASELINE configs[1] / [4] at N > 1; SURVEY.md 8e, last sentence): the MCU loop of
    encoder::encode (ref encoder/jpezy_encoder.hpp:55-67) is separable by MCU rows for everything but pre_DC and the bit
    cursor, which belong to the Huffman tail -- so the unit that is sharded is the MCU row, rank k transforms the contiguous
    rows shard_range(mcu_rows, world, k), and the coefficient rows are gathered on `root` band by band, overlapped with the
    transform of the next band (gather_to_root_pipelined with MCU rows in the place of frames).  The Huffman stage then runs
    on `root` over the whole frame (host or GPU coder): its DC predictors and bit cursor never cross a rank.

    encode_band(mcu_lo, mcu_hi, dst): enqueue the transform of MCU rows [mcu_lo, mcu_hi) -- pixel rows band_pixel_rows(...) of
        the planes, as a frame W x that many rows -- into dst, an int16 tensor [mcu_hi - mcu_lo, mcu_cols * (4|6) * 64].
    Returns the frame's coefficients [mcu_rows, mcu_cols * (4|6) * 64] on root (the layout of a whole-frame call), None elsewhere.
    """
    mcu_cols, mcu_rows = (W + 15) // 16, (H + 15) // 16
    per_row = mcu_cols * (4 if gray else 6) * 64
    return gather_to_root_pipelined(encode_band, mcu_rows, per_row, band_rows, device, root=root, group=group, out=out, ring=ring)


class JpgBatch:
    """What gather_jpg_to_root_pipelined leaves on the consumer: the batch's .jpg files, packed chunk by chunk.
    sizes: int64 CPU tensor [n_frames]; chunks: list of (lo, hi, packed uint8 tensor) in frame order -- frame f of a chunk
    starts at the sum of the sizes of the chunk's earlier frames."""

    def __init__(self, n_frames):
        self.sizes = torch.zeros(n_frames, dtype=torch.int64)
        self.chunks = []

    def finish(self):
        self.chunks.sort(key=lambda c: c[0])
        return self

    def frame(self, f):
        for lo, hi, packed in self.chunks:
            if lo <= f < hi:
                off = int(self.sizes[lo:f].sum())
                return packed[off: off + int(self.sizes[f])]
        raise IndexError(f)

    def total_bytes(self):
        return int(self.sizes.sum())

    def equals(self, other):
        """same sizes and the same bytes in frame order, whatever the chunking of the two batches"""
        if not torch.equal(self.sizes, other.sizes):
            return False
        a = [c[2] for c in self.chunks if c[2].numel()]
        b = [c[2] for c in other.chunks if c[2].numel()]
        if not a or not b:
            return not a and not b
        dev = a[0].device
        # walk both chunk lists side by side: compare the overlapping byte ranges (no copy of the whole batch)
        ia = ib = oa = ob = 0
        while ia < len(a) and ib < len(b):
            n = min(a[ia].numel() - oa, b[ib].numel() - ob)
            if not torch.equal(a[ia][oa:oa + n], b[ib][ob:ob + n].to(dev)):
                return False
            oa += n
            ob += n
            if oa == a[ia].numel():
                ia, oa = ia + 1, 0
            if ob == b[ib].numel():
                ib, ob = ib + 1, 0
        return ia == len(a) and ib == len(b)


def gather_jpg_to_root_pipelined(encode_chunk, n_frames, chunk_frames, device, root=0, group=None, meta_group=None, ring=3,
                                 solo=False):
    """Sharded END-TO-END encode: every rank runs FDCT+quantise AND the Huffman stage on its own frames; the finished
    .jpg files -- variable length -- are gathered on `root`, chunk by chunk, overlapped with the encoding of the next chunk.

    encode_chunk(lo, hi, slot) -> (buf, sizes): enqueue (on the CURRENT stream of `device`, or compute synchronously on the
        CPU) the encode of global frames [lo, hi) -- inside this rank's shard -- and return buf, a uint8 tensor
        [hi - lo, stride] holding frame k's file in buf[k, :sizes[k]], and sizes, an int64 tensor [hi - lo] (a negative
        size is that frame's error code: raised here).  slot in 0..ring-1 names the staging buffers the callee may use; a
        slot is handed out again only after the chunk that used it has been packed (stream-ordered on the GPU).
    Protocol per chunk and sender: its sizes (fixed length: chunk_frames int64, through meta_group -- a gloo group, host to
        host -- when given, else through `group` as a device tensor), then its files packed back to back (one message of
        exactly sum(sizes) bytes; both sides know the length from the sizes).  A chunk is packed and sent while the next one is
        being encoded (the encode of chunk c + 1 is enqueued before chunk c is waited for).
    solo: this process alone codes ALL frames through the same chunking, packing and staging ring, whatever process group
        exists (the one-GPU reference of the same pipeline, measured inside a multi-rank job).
    Returns a JpgBatch on `root`, None elsewhere.
    """
    multi = not solo and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    world = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    on_gpu = torch.device(device).type == "cuda"
    spans = [chunk_spans(*shard_range(n_frames, world, r), chunk_frames) for r in range(world)]
    rounds = max((len(s) for s in spans), default=0)
    if solo:
        root = rank
    result = JpgBatch(n_frames) if rank == root else None
    comm = torch.cuda.Stream(device) if on_gpu else None
    slot_free = [None] * ring            # GPU: event after the packing copy that last read the slot's staging buffers
    inflight = {}                        # chunk index -> (lo, hi, buf, sizes, event)
    works, keep = [], []                 # outstanding sends / receives and the tensors they still use

    def enqueue(c):
        lo, hi = spans[rank][c]
        slot = c % ring
        if on_gpu and slot_free[slot] is not None:
            torch.cuda.current_stream(device).wait_event(slot_free[slot])
        buf, sizes = encode_chunk(lo, hi, slot)
        ev = None
        if on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
        inflight[c] = (lo, hi, buf, sizes, ev)

    def pack(c):
        """-> (lo, hi, sizes on the host, the chunk's files back to back)"""
        lo, hi, buf, sizes, ev = inflight.pop(c)
        if on_gpu:
            comm.wait_event(ev)
            with torch.cuda.stream(comm):
                hs = sizes.to("cpu", non_blocking=False)          # waits for this chunk's kernels only
                if int(hs.min()) < 0:
                    raise RuntimeError(f"frames {lo}..{hi}: encode failed with status {int(hs.min())}")
                packed = torch.cat([buf[k, : int(hs[k])] for k in range(hi - lo)]) if hi > lo else buf.new_empty(0)
                done = torch.cuda.Event()
                done.record(comm)
            slot_free[c % ring] = done
        else:
            hs = sizes.clone()
            if int(hs.min()) < 0:
                raise RuntimeError(f"frames {lo}..{hi}: encode failed with status {int(hs.min())}")
            packed = torch.cat([buf[k, : int(hs[k])] for k in range(hi - lo)]) if hi > lo else buf.new_empty(0)
        return lo, hi, hs, packed

    def send_sizes(hs):
        pad = torch.zeros(chunk_frames, dtype=torch.int64)
        pad[: hs.numel()] = hs
        if meta_group is not None:
            dist.send(pad, root, group=meta_group)
        else:
            t = pad.to(device) if on_gpu else pad
            keep.append(t)
            if on_gpu:
                with torch.cuda.stream(comm):
                    works.append(dist.isend(t, root, group=group))
            else:
                works.append(dist.isend(t, root, group=group))

    def recv_sizes(senders):
        """{r: sizes of r's current chunk} -- host to host through meta_group, else one batch of device receives"""
        pads = {r: torch.zeros(chunk_frames, dtype=torch.int64) for r in senders}
        if meta_group is not None:
            for r in senders:
                dist.recv(pads[r], r, group=meta_group)
            return pads
        if on_gpu:
            dev = {r: pads[r].to(device) for r in senders}
            with torch.cuda.stream(comm):
                for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, dev[r], r, group) for r in senders]):
                    w.wait()
                return {r: dev[r].cpu() for r in senders}
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, pads[r], r, group) for r in senders]):
            w.wait()
        return pads

    def transfer(c):
        if rank == root:
            if c < len(spans[rank]):
                lo, hi, hs, packed = pack(c)
                result.sizes[lo:hi] = hs
                result.chunks.append((lo, hi, packed))
            senders = [r for r in range(world) if r != root and c < len(spans[r])] if multi else []
            if not senders:
                return
            got = recv_sizes(senders)
            ops = []
            for r in senders:
                lo, hi = spans[r][c]
                hs = got[r][: hi - lo]
                result.sizes[lo:hi] = hs
                total = int(hs.sum())
                if on_gpu:
                    with torch.cuda.stream(comm):
                        dst = torch.empty(total, dtype=torch.uint8, device=device)
                else:
                    dst = torch.empty(total, dtype=torch.uint8)
                if total:
                    ops.append(dist.P2POp(dist.irecv, dst, r, group))
                result.chunks.append((lo, hi, dst))
            if ops:       # one group: the senders' files come in side by side, one xGMI link each
                if on_gpu:
                    with torch.cuda.stream(comm):
                        works.extend(dist.batch_isend_irecv(ops))
                else:
                    works.extend(dist.batch_isend_irecv(ops))
        elif c < len(spans[rank]):
            lo, hi, hs, packed = pack(c)
            send_sizes(hs)
            if packed.numel():
                keep.append(packed)
                if on_gpu:
                    with torch.cuda.stream(comm):
                        works.append(dist.isend(packed, root, group=group))
                else:
                    works.append(dist.isend(packed, root, group=group))

    for c in range(rounds + 1):
        if c < len(spans[rank]):
            enqueue(c)                   # chunk c is being encoded ...
        if c >= 1:
            transfer(c - 1)              # ... while chunk c - 1 is packed and travels
    for w in works:
        w.wait()
    if on_gpu:
        torch.cuda.current_stream(device).wait_stream(comm)
    return result.finish() if rank == root else None
