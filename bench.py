#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the fused encode (FDCT+quant) stage on MI355X, with its HBM roofline
fraction and the CPU oracle timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A *step* is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
the default workload is BASELINE.json configs[1], one 4096x4096 random-pixel frame per step.  Steps walk
a ring of distinct frames whose inputs+outputs exceed the 256 MiB Infinity Cache, so the kernel really
streams from HBM.  The K timed steps are a fixed sequence of K launches (one per step, on torch's
current stream): they are captured once into a hipGraph; a *replay* of that graph is exactly K steps,
bracketed by barrier + synchronize on both sides.  The replay is repeated R times (--repeats, default 11);
`ms_per_step` is the MEDIAN replay divided by K (max over ranks per replay), min/max are reported beside it; the W warm-up
steps are run again before every repetition's bracket (--rewarm 1), so that each repetition is the contract's own sequence.

--gpus N > 1: when WORLD_SIZE is not set this process starts N copies of itself, one per GPU (RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) *before it touches the GPU*, and rank 0 prints the
JSON line; under torch.distributed.run the ranks are taken from the environment as usual.  `value` at N > 1
is the same workload on every rank (frames are independent: no data-path collective, weak scaling).
In addition every N > 1 run (and N = 1 with --batch) measures BASELINE configs[3] as north_star states it --
a batch of 4096 1920x1080 frames sharded by jpezy_amd.sharding.shard_range, strong scaling -- and reports it
as the object "batch": kernel-only throughput, the RCCL gather of the coefficient buffers to the consumer
rank, the pipelined end-to-end time (gather overlapped with the kernels chunk by chunk), and the speed-up
over the same pipeline run on ONE GPU in the same job.

Prints ONE JSON line (rank 0).  `oracle/` is used only for the cpu_baseline leg.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# multi-process GPU work on this pool needs dmabuf IPC (RCCL across ranks fails with the legacy mode); set before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (W, H, gray, frames_per_step, direction, description)
    "encode4096": (4096, 4096, False, 1, "encode", "BASELINE configs[1]: single 4096x4096 random-pixel frame encode"),
    "decode4096": (4096, 4096, False, 1, "decode", "BASELINE configs[2]: single 4096x4096 decode (dequant+IDCT+RGB)"),
    "gray8k": (7680, 4320, True, 1, "encode", "BASELINE configs[4]: 7680x4320 --gray encode"),
    "batch1080p": (1920, 1080, False, 32, "encode", "BASELINE configs[3] shard: 32 frames 1920x1080 per step"),
    "gray8k_decode": (7680, 4320, True, 1, "decode", "BASELINE configs[4]: 7680x4320 --gray decode (r = g = b = Y)"),
    "encode4096_jpg": (4096, 4096, False, 1, "encode+entropy",
                       "4096x4096 planes -> complete .jpg in HBM (FDCT+quant kernel, then the GPU Huffman stage)"),
    "decode4096_jpg": (4096, 4096, False, 1, "entropy+decode",
                       "4096x4096 .jpg bytes (host) -> GPU Huffman decoder -> dequant+IDCT+RGB planes in HBM"),
}
BATCH_W, BATCH_H, BATCH_FRAMES = 1920, 1080, 4096      # BASELINE configs[3]


def algorithmic_bytes(W, H, gray, direction):
    """encode colour 3 B/px read + 1.5 samples x 2 B written; gray 3 + 2; decode colour
    3 B/px of coefficients read + 3 B/px written; decode gray 2 + 3: the kernel does not read the chroma blocks of a
    GRAY_MODE decode (r = g = b = clamp(Y), ref decoder/jpezy_decoder.hpp:561), so they are not in the numerator either --
    a fraction's numerator must not exceed what the launch moves (the counters say 166.2 MB for 8K).
    Padded MCU grid for the coefficient side."""
    mc, mr = (W + 15) // 16, (H + 15) // 16
    px = W * H
    if direction.startswith("encode"):
        return 3 * px + mc * mr * (4 if gray else 6) * 128
    return mc * mr * (4 if gray else 6) * 128 + 3 * px


# ------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves, before any GPU call in this process
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv, timeout=None):
    """Start `n` copies of the command `argv` (one per rank: RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE,
    MASTER_ADDR=127.0.0.1, MASTER_PORT in their environment) and wait for them.  Rank 0 inherits stdout (its one JSON
    line is the job's line); the other ranks' stdout is dropped, stderr is shared.  Returns the first non-zero exit
    code (0 if every rank succeeded); when a rank fails the others are terminated by PID.  The caller must not have
    initialised the GPU: the children are plain child processes, nothing is exec'ed over a live HIP context."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc, t0 = 0, time.time()
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for q in live:
                    procs[q].terminate()
        if timeout is not None and time.time() - t0 > timeout and live:
            print("bench.py: ranks timed out", file=sys.stderr)
            for q in live:
                procs[q].terminate()
            rc = rc or 124
            timeout = None
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (scalar C port of the reference) on the box's host cores
# ------------------------------------------------------------------------------------------------------------------
def _host_cores():
    """cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a job a
    share of the host -- 16 cores per GPU -- through the quota while the mask still shows every core)"""
    try:
        n = max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]            # cgroup v2
        if q != "max":
            quota = int(q) / int(per)
    except Exception:
        try:
            q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())        # cgroup v1
            per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0:
                quota = q / per
        except Exception:
            quota = None
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def _timed_bands(fn, bands, passes, threads):
    """run fn(y0, y1) over every band `passes` times on `threads` host threads (the C calls release the GIL)"""
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    if threads == 1:
        for _ in range(passes):
            for y0, y1 in bands:
                fn(y0, y1)
    else:
        def work(band):
            for _ in range(passes):
                fn(*band)
        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(work, bands))
    return time.perf_counter() - t0


def cpu_baseline(W, H, gray, direction, budget_s=8.0):
    """the oracle's stages on the host, (i) one thread -- the reference is single-threaded --
    and (ii) MCU-row bands over all host cores, for the compute stage the GPU kernel of this workload replaces
    (encode: colour+subsample+FDCT+quant+zig-zag; decode: dequant+IDCT+upsample+colour), plus the Huffman stage
    (tail or head) and the resulting whole-codec rate on one thread.  A bounded band of the workload's frame."""
    import ctypes as C
    import numpy as np
    from oracle import oracle as O
    L = O.lib()
    cores = _host_cores()
    enc = direction.startswith("encode")
    Hb = min(H, 1024) // 16 * 16 or 16                     # band frame: W x Hb, same width => same memory pattern per row
    mr = Hb // 16
    r, g, b = O.synth_rgb(W, min(Hb, 256))
    reps = -(-Hb // min(Hb, 256))
    r, g, b = (np.ascontiguousarray(np.tile(p, reps)[: W * Hb]) for p in (r, g, b))
    mc = (W + 15) // 16
    u8, i16 = (lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))), (lambda a: a.ctypes.data_as(C.POINTER(C.c_int16)))
    if enc:
        co = np.zeros((mr, mc, 4 if gray else 6, 64), dtype=np.int16)
        fn = lambda y0, y1: L.jo_encode_coeffs_rows(u8(r), u8(g), u8(b), W, Hb, int(gray), y0, y1, i16(co))  # noqa: E731
        stage = "colour+subsample+FDCT+quant+zig-zag (ref encoder/jpezy_encoder.hpp:90-172)"
    else:
        co = np.zeros((mr, mc, 6, 64), dtype=np.int16)
        L.jo_encode_coeffs_rows(u8(r), u8(g), u8(b), W, Hb, 0, 0, mr, i16(co))       # real coefficients of the band
        info = O.make_info(W, Hb)
        outp = [np.zeros(W * Hb, dtype=np.uint8) for _ in range(3)]
        fn = lambda y0, y1: L.jo_decode_planes_rows(i16(co), C.byref(info), int(gray), y0, y1, u8(outp[0]), u8(outp[1]), u8(outp[2]))  # noqa: E731
        stage = "dequant+IDCT+upsample+YCbCr->RGB (ref decoder/jpezy_decoder.hpp:504-578,645-676)"
    t0 = time.perf_counter()
    fn(0, 1)
    per_row = max(time.perf_counter() - t0, 1e-6)
    # (i) one thread
    rows1 = max(1, min(mr, int(budget_s / per_row)))
    passes1 = max(1, int(budget_s / (per_row * rows1)))
    dt1 = _timed_bands(fn, [(0, rows1)], passes1, 1)
    px1 = rows1 * 16 * W * passes1
    # (ii) all cores: the band's MCU rows dealt to `cores` threads, each its own contiguous rows
    per = max(1, mr // cores)
    bands = [(y, min(y + per, mr)) for y in range(0, per * cores, per) if y < mr][:cores]
    passesN = max(1, int(0.5 * budget_s / (per_row * per)))
    dtN = _timed_bands(fn, bands, passesN, len(bands))
    pxN = sum(y1 - y0 for y0, y1 in bands) * 16 * W * passesN
    out = {
        "value": round(px1 / dt1 / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port", "stage": stage,
        "sample": f"{passes1} pass(es) over {rows1} MCU rows ({px1 / 1e6:.1f} Mpx) of a {W}x{Hb} band of the workload's frame, "
                  f"oracle/jpezy_oracle.c -O2 -ffp-contract=off, {dt1:.1f} s",
        "all_cores": {"value": round(pxN / dtN / 1e6, 3), "unit": "Mpixels/s", "cores": len(bands),
                      "sample": f"{len(bands)} threads x {passesN} pass(es) over {per} MCU rows each ({pxN / 1e6:.1f} Mpx), {dtN:.1f} s"},
    }
    # Huffman stage on one thread (compute / Huffman / total split).  Whole band.
    try:
        t0 = time.perf_counter()
        jpg = O.write_jpeg(co, W, Hb, gray=bool(gray and enc))
        t_tail = time.perf_counter() - t0
        if enc:
            t_h, name = t_tail, "encode_huffman + jpezy_writer (ref encoder/jpezy_encoder.hpp:174-225)"
        else:
            t0 = time.perf_counter()
            O.read_jpeg(jpg)
            t_h, name = time.perf_counter() - t0, "marker parser + decode_huffman (ref decoder/jpezy_decoder.hpp:171-502,583-642)"
        pxb = W * Hb
        t_c = pxb / (px1 / dt1)
        out["huffman_stage"] = {"value": round(pxb / t_h / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "what": name,
                                "sample": f"one {W}x{Hb} band ({pxb / 1e6:.1f} Mpx, {len(jpg)} B of .jpg), {t_h:.2f} s"}
        out["total_1core"] = {"value": round(pxb / (t_c + t_h) / 1e6, 3), "unit": "Mpixels/s",
                              "note": "compute stage + Huffman stage, one thread (what jpezy's own MCU loop does)"}
    except Exception as e:      # the baseline must never break the bench line
        out["huffman_stage"] = {"error": str(e)}
    return out


# ------------------------------------------------------------------------------------------------------------------
def synth_frames(torch, lo, hi, plane, dev, seed=0x6A70657A79):
    """planes r,g,b [hi-lo, plane] u8 of global frames [lo, hi): frame f is drawn from its own generator (seed + f),
    so a frame has the same pixels whichever rank or shard generates it"""
    n = hi - lo
    out = torch.empty((3, max(n, 0), plane), dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    for k in range(n):
        gen.manual_seed(seed + lo + k)
        out[:, k] = torch.randint(0, 256, (3, plane), dtype=torch.uint8, device=dev, generator=gen)
    return out[0], out[1], out[2]


def run_batch(args, torch, dist, J, ctx, dev, rank, world, multi, comm_cpu):
    """BASELINE configs[3]: `--batch-frames` 1920x1080 frames sharded over the ranks (strong scaling).  See the
    module docstring for what is reported.  comm_cpu: rehearsal on one GPU (gloo cannot move device tensors)."""
    from jpezy_amd import sharding
    if os.environ.get("JPEZY_BENCH_FAIL_BATCH"):          # tests/test_gpu_parity.py: a failing batch leg must not cost the headline line
        raise RuntimeError("injected failure (JPEZY_BENCH_FAIL_BATCH)")
    W, H, F, chunk = BATCH_W, BATCH_H, args.batch_frames, args.batch_chunk
    plane = W * H
    cpf = J.coeff_count(W, H, False)
    lo, hi = sharding.shard_range(F, world, rank)
    n_local = hi - lo
    stream = torch.cuda.current_stream(dev)
    pr, pg, pb = synth_frames(torch, lo, hi, plane, dev)
    local = torch.empty((max(n_local, 1), cpf), dtype=torch.int16, device=dev)

    def kernels(frames_lo, frames_hi, pr_, pg_, pb_, base, dst):
        """frames [frames_lo, frames_hi) (global) whose planes start at global frame `base` in pr_/pg_/pb_"""
        a, b = frames_lo - base, frames_hi - base
        ctx.fdct_quant_dev(pr_[a:b], pg_[a:b], pb_[a:b], W, H, dst, gray=False, n_frames=b - a, plane_stride=plane,
                           stream=torch.cuda.current_stream(dev).cuda_stream)

    def sync_all():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()

    def timed(fn, reps=3):
        ts = []
        for _ in range(reps):
            sync_all()
            t0 = time.perf_counter()
            fn()
            sync_all()
            ts.append(time.perf_counter() - t0)
        red = torch.tensor(ts, dtype=torch.float64, device="cpu" if comm_cpu else dev)
        if multi:
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
        return statistics.median(red.tolist())

    # (1) kernel-only: every rank encodes its shard, chunk by chunk, into its local buffer
    def kernel_pass():
        for a, b in sharding.chunk_spans(lo, hi, chunk):
            kernels(a, b, pr, pg, pb, lo, local[a - lo:b - lo])
    kernel_pass()
    t_kernel = timed(kernel_pass)

    # (2) + (3) gather alone (chunks are copies of the finished local buffer) and the pipelined end-to-end pass
    comm_dev = "cpu" if comm_cpu else dev
    out = None
    if rank == 0:
        out = torch.empty((F, cpf), dtype=torch.int16, device=comm_dev)

    def chunk_from_local(a, b, dst):
        if comm_cpu:
            dst.copy_(local[a - lo:b - lo])
        else:
            dst.copy_(local[a - lo:b - lo], non_blocking=True)

    def chunk_encode(a, b, dst):
        if comm_cpu:
            kernels(a, b, pr, pg, pb, lo, local[a - lo:b - lo])
            torch.cuda.synchronize(dev)
            dst.copy_(local[a - lo:b - lo])
        else:
            kernels(a, b, pr, pg, pb, lo, dst)

    def gather_pass():
        sharding.gather_to_root_pipelined(chunk_from_local, F, cpf, chunk, comm_dev, root=0, out=out)

    def e2e_pass():
        sharding.gather_to_root_pipelined(chunk_encode, F, cpf, chunk, comm_dev, root=0, out=out)

    t_gather = t_e2e = None
    if multi:
        gather_pass()                       # first use builds the point-to-point connections
        t_gather = timed(gather_pass)
        e2e_pass()
        t_e2e = timed(e2e_pass)
    else:
        t_e2e = t_kernel                    # one GPU: nothing to gather, the consumer already holds the batch

    # (3b) the pipeline that is not capped by the consumer's xGMI links: encoder::encode END TO END on every rank (FDCT+quant,
    #      then the GPU Huffman stage: ref encoder/jpezy_encoder.hpp:38-77) and only the finished .jpg files gathered
    jstride = args.batch_jpg_stride
    jring = 3
    jbuf = [torch.empty((chunk, jstride), dtype=torch.uint8, device=dev) for _ in range(jring)]
    jsz = [torch.zeros(chunk, dtype=torch.int64, device=dev) for _ in range(jring)]
    jco = torch.empty((chunk, cpf), dtype=torch.int16, device=dev)
    meta = dist.new_group(backend="gloo") if (multi and not comm_cpu) else None      # the sizes travel host to host

    def jpg_chunk_from(pr_, pg_, pb_, base):
        def fn(a, b, slot):
            n = b - a
            kernels(a, b, pr_, pg_, pb_, base, jco[:n])
            ctx.write_jpeg_gpu_dev(jco[:n], W, H, jbuf[slot][:n], jsz[slot][:n], n_frames=n,
                                   stream=torch.cuda.current_stream(dev).cuda_stream)
            if comm_cpu:
                torch.cuda.synchronize(dev)
                return jbuf[slot][:n].cpu(), jsz[slot][:n].cpu()
            return jbuf[slot][:n], jsz[slot][:n]
        return fn

    jpg_res = [None]

    def e2e_jpg_pass():
        jpg_res[0] = sharding.gather_jpg_to_root_pipelined(jpg_chunk_from(pr, pg, pb, lo), F, chunk, comm_dev, root=0,
                                                           meta_group=meta, ring=jring)
    e2e_jpg_pass()                           # first use: entropy scratch, header cache, point-to-point connections
    t_e2e_jpg = timed(e2e_jpg_pass)

    # (4) the same pipelines on ONE GPU, in the same job (rank 0 alone; the others wait), and the parity property of the
    #     sharded paths: the gathered batch equals the single-GPU batch bit for bit
    t_one, same = None, None
    t_one_jpg, same_jpg, jpg_total = None, None, None
    if rank == 0 and not multi:
        t_one_jpg, jpg_total = t_e2e_jpg, jpg_res[0].total_bytes()
    if multi:
        if rank == 0:
            fr, fg, fb = synth_frames(torch, 0, F, plane, dev)
            full = torch.empty((F, cpf), dtype=torch.int16, device=dev)

            def one_pass():
                for a, b in sharding.chunk_spans(0, F, chunk):
                    kernels(a, b, fr, fg, fb, 0, full[a:b])
            one_pass()
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                one_pass()
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter() - t0)
            t_one = statistics.median(ts)
            same = bool(torch.equal(out.to(dev) if comm_cpu else out, full))
            del full
            solo = [None]

            def one_jpg_pass():
                solo[0] = sharding.gather_jpg_to_root_pipelined(jpg_chunk_from(fr, fg, fb, 0), F, chunk, comm_dev, ring=jring, solo=True)
            one_jpg_pass()
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                one_jpg_pass()
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter() - t0)
            t_one_jpg = statistics.median(ts)
            same_jpg = bool(jpg_res[0].equals(solo[0]))
            jpg_total = solo[0].total_bytes()
            del fr, fg, fb, solo
        dist.barrier()
    else:
        t_one = t_kernel

    if rank != 0:
        return None
    px = F * plane
    bytes_moved = (F - (sharding.shard_range(F, world, 0)[1])) * cpf * 2 if multi else 0
    res = {
        "workload": f"BASELINE configs[3]: batch of {F} frames {W}x{H}, sharded by shard_range over {world} GPU(s), strong scaling",
        "frames": F, "frames_per_rank": sharding.max_shard(F, world), "chunk_frames": chunk,
        "kernel_only": {"ms": round(t_kernel * 1e3, 3), "Mpixels_per_s": round(px / t_kernel / 1e6, 1),
                        "note": "every rank encodes its shard; max over ranks; median of 3 passes"},
        "end_to_end": {"ms": round(t_e2e * 1e3, 3), "Mpixels_per_s": round(px / t_e2e / 1e6, 1),
                       "note": "inputs resident in each rank's HBM -> whole batch of coefficients on rank 0; gather overlapped with the kernels"},
        "one_gpu_same_pipeline": {"ms": round(t_one * 1e3, 3), "Mpixels_per_s": round(px / t_one / 1e6, 1),
                                  "note": "all frames on rank 0's GPU, same chunking, measured in this job"},
        "speedup_kernel_only": round(t_one / t_kernel, 3),
        "speedup_end_to_end": round(t_one / t_e2e, 3),
        "end_to_end_jpg": {"ms": round(t_e2e_jpg * 1e3, 3), "Mpixels_per_s": round(px / t_e2e_jpg / 1e6, 1),
                           "jpg_bytes": jpg_total, "out_stride": jstride,
                           "note": "encoder::encode end to end on every rank (FDCT+quant, then the GPU Huffman stage); only the "
                                   "finished .jpg files travel to rank 0 (sizes host to host, files packed per chunk, overlapped with "
                                   "the next chunk's kernels)"},
        "one_gpu_same_pipeline_jpg": {"ms": round(t_one_jpg * 1e3, 3), "Mpixels_per_s": round(px / t_one_jpg / 1e6, 1),
                                      "note": "all frames on rank 0's GPU through the same chunked FDCT + Huffman + packing pipeline"},
        "speedup_end_to_end_jpg": round(t_one_jpg / t_e2e_jpg, 3),
    }
    if multi:
        res["gather"] = {"op": "point-to-point sends to rank 0 (batch_isend_irecv), chunks of the finished local buffers",
                         "ms": round(t_gather * 1e3, 3), "bytes_into_rank0": bytes_moved,
                         "GBs_into_rank0": round(bytes_moved / t_gather / 1e9, 2)}
        hidden = t_kernel + t_gather - t_e2e
        res["overlap_efficiency"] = round(max(0.0, min(1.0, hidden / max(min(t_kernel, t_gather), 1e-12))), 3)
        res["gathered_equals_one_gpu_result"] = same
        res["gathered_jpg_equals_one_gpu_result"] = same_jpg
        own = int(jpg_res[0].sizes[slice(*sharding.shard_range(F, world, 0))].sum())
        res["end_to_end_jpg"]["bytes_into_rank0"] = jpg_total - own
        if comm_cpu:
            res["rehearsal"] = "all ranks on GPU 0, gloo + host staging: control flow only, the numbers mean nothing"
    return res



# ------------------------------------------------------------------------------------------------------------------
# The other single-GPU BASELINE configs on the driver's default line: configs[2] (decode 4096^2, bit-exact and tolerant) and
# configs[4] (7680x4320 --gray encode and decode), a few replays each AFTER the timed region, reported beside `value`, never in it
# ------------------------------------------------------------------------------------------------------------------
OTHER_KERNELS = {"encode": "f32::fdct_quant_f32_kernel", "decode": "dequant_idct_kernel"}


def measure_other_workloads(torch, J, ctx, dev, steps=40, replays=3, copy_gbs=None):
    """Every entry: the same protocol as the headline in small -- a ring of distinct frames larger than the Infinity Cache,
    `steps` launches captured as one hipGraph, the median of `replays` replays timed with HIP events on the launch stream."""
    traffic = {}
    tfile = ROOT / "profiles" / "traffic.json"
    if tfile.exists():
        try:
            traffic = json.loads(tfile.read_text())
        except Exception:
            traffic = {}
    out = {}
    stream = torch.cuda.Stream(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x6A70657A7A)
    for name, tolerant in (("decode4096", False), ("decode4096", True), ("gray8k", False), ("gray8k_decode", False)):
        key = name + ("_tolerant" if tolerant else "")
        try:
            W, H, gray, fps, direction, desc = WORKLOADS[name]
            plane = W * H
            ncoef = J.coeff_count(W, H, gray if direction == "encode" else False)
            step_bytes = algorithmic_bytes(W, H, gray, direction) * fps
            ring = max(2, -(-(512 << 20) // step_bytes))
            pr, pg, pb = (torch.randint(0, 256, (ring, plane), dtype=torch.uint8, device=dev, generator=gen) for _ in range(3))
            co = torch.empty((ring, ncoef), dtype=torch.int16, device=dev)
            with torch.cuda.stream(stream):
                if direction == "decode":      # real coefficients: the random frames encoded once (6-block layout)
                    for k in range(ring):
                        ctx.fdct_quant_dev(pr[k], pg[k], pb[k], W, H, co[k], gray=False, stream=stream.cuda_stream)

                    def step(i):
                        k = i % ring
                        ctx.dequant_idct_dev(co[k], W, H, pr[k], pg[k], pb[k], gray=gray, stream=stream.cuda_stream)
                else:
                    def step(i):
                        k = i % ring
                        ctx.fdct_quant_dev(pr[k], pg[k], pb[k], W, H, co[k], gray=gray, stream=stream.cuda_stream)
                ctx.set_decode_tolerance(1 if tolerant else 0)
                for i in range(4):
                    step(i)
                stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    for i in range(steps):
                        step(i)
                stream.synchronize()
                g.replay()
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ms = []
                for _ in range(replays):
                    e0.record(stream)
                    g.replay()
                    e1.record(stream)
                    stream.synchronize()
                    ms.append(e0.elapsed_time(e1) / steps)
            m = statistics.median(ms)
            gbs = step_bytes / (m * 1e-3) / 1e9
            tj = traffic.get(name, {}) if not tolerant else {}
            out[key] = {"workload": desc + (" -- opt-in tolerance mode (within 1 LSB)" if tolerant else ""),
                        "ms_per_step": round(m, 5), "value": round(plane * fps / (m * 1e-3) / 1e6, 2), "unit": "Mpixels/s",
                        "steps": steps, "replays": replays, "kernel": OTHER_KERNELS[direction],
                        "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": step_bytes,
                                     "traffic": tj.get("bytes_per_launch"), "traffic_source": tj.get("source"),
                                     "device_copy_GBs_measured": round(copy_gbs, 1) if copy_gbs else None,
                                     "frac_of_device_copy": round(gbs / copy_gbs, 4) if copy_gbs else None}}
            del g, pr, pg, pb, co
        except Exception as e:          # never at the cost of the headline line
            out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            ctx.set_decode_tolerance(0)
    ctx.fallback_count()
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------------------------
# The native multi-GPU entry of the C-ABI (jpezy_multi_create / jpezy_multi_encode: ONE host process, a lane per device -- pinned
# staging rings, feeder / drainer threads, gather by peer copies), timed by rank 0 over the GPUs of the job after everything else.
# Host planes in, .jpg files out in host memory: PCIe-inclusive, reported beside `value`, never in it.  The handle is created outside
# the bracket (that is what it is for), the output buffers are touched before it, nothing but the C call is inside.  Reference for
# the link: a pinned hipMemcpy host -> device measured in this same run.  On a one-GPU run it also runs two lanes on the one device
# (the multi-lane path), and the round-5 one-shot shape (16 frames, handle built and torn down inside the call) beside it.
# ------------------------------------------------------------------------------------------------------------------
def _pinned_h2d_GBs(torch, dev, nbytes=256 << 20, reps=5):
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    h.fill_(7)
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(dev)
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        d.copy_(h, non_blocking=True)
        e1.record()
        e1.synchronize()
        best = max(best, nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del h, d
    return best


def measure_native_multi(J, ctx, n_dev, frames_per_dev=256, max_frames=1024):
    import ctypes as C

    import numpy as np
    import torch
    from jpezy_amd import api
    lib = api.load_library()
    W, H = BATCH_W, BATCH_H
    plane = W * H
    rng = np.random.default_rng(0x6A70)
    base = [rng.integers(0, 256, (4, plane), dtype=np.uint8) for _ in range(3)]          # four distinct frames, repeated
    stride = 1 << 20                                                                      # a 1080p random-pixel frame codes to ~0.66 MB
    comment = b"Encoded by jpezy"
    pcie = _pinned_h2d_GBs(torch, torch.device("cuda", 0))
    res = {"workload": f"{W}x{H} frames, host planes -> .jpg files in host memory (PCIe inclusive), jpezy_multi_encode on a handle created "
                       "outside the bracket; planes are pageable numpy memory (staged through the lanes' pinned rings) unless a layout says pinned",
           "entry": "include/jpezy_hip.h: jpezy_multi_create / jpezy_multi_encode (one host process; per device a context, a ring of 6 pinned + "
                    "device slots, feeder threads as the host's cores allow (2..4 per device) and 2 drainer threads (as many as feeders when coefficients go back to host memory))",
           "pcie_h2d_GBs_pinned_hipMemcpy": round(pcie, 1),
           "never_run_on_two_different_gpus_by_the_builder": True}
    ref = ctx.encode_jpeg(base[0][1], base[1][1], base[2][1], W, H)

    def planes_for(F):
        return [np.ascontiguousarray(np.tile(b, (F // 4 + 1, 1))[:F]).reshape(-1) for b in base]

    def run(name, devs, F, planes, chunk_frames=0, reps=3):
        jpg = np.zeros(F * stride, dtype=np.uint8)                                        # touched: no first-touch faults inside the bracket
        sizes = (C.c_longlong * F)()
        out = api.MultiOut()
        out.jpg, out.jpg_stride, out.jpg_sizes, out.on_root_device = jpg.ctypes.data, stride, sizes, 0
        ptrs = [C.c_void_p(p.data_ptr()) if hasattr(p, "data_ptr") else api._np_ptr(p) for p in planes]
        darr = (C.c_int * len(devs))(*devs)
        h = lib.jpezy_multi_create(darr, len(devs), W, H, 0, chunk_frames)
        if not h:
            raise RuntimeError(lib.jpezy_hip_last_error().decode(errors="replace"))
        try:
            times = []
            for i in range(reps + 1):                                                     # first call: output-dependent buffers, first touch
                t0 = time.perf_counter()
                rc = lib.jpezy_multi_encode(h, *ptrs, F, comment, C.byref(out))
                dt = time.perf_counter() - t0
                if rc != 0:
                    raise RuntimeError(lib.jpezy_hip_last_error().decode(errors="replace"))
                if i:
                    times.append(dt)
            dt = statistics.median(times)
            st = (api.MultiLaneStats * len(devs))()
            lib.jpezy_multi_last_stats(h, st, len(devs))
            chunk = lib.jpezy_multi_chunk_frames(h)
            feeders = lib.jpezy_multi_feeder_threads(h)
        finally:
            lib.jpezy_multi_destroy(h)
        ok = all(sizes[f] == len(ref) and jpg[f * stride: f * stride + sizes[f]].tobytes() == ref for f in (1, F - 3))   # copies of base frame 1
        up = 3 * plane * F
        per_dev_gbs = up / len(set(devs)) / dt / 1e9
        res[name] = {"devices": devs, "frames": F, "chunk_frames": chunk, "feeder_threads_per_lane": feeders, "calls_timed": reps, "ms": round(dt * 1e3, 2),
                     "ms_min_max": [round(min(times) * 1e3, 2), round(max(times) * 1e3, 2)],
                     "Mpixels_per_s": round(F * plane / dt / 1e6, 1), "GBs_h2d": round(up / dt / 1e9, 2),
                     "GBs_h2d_per_device": round(per_dev_gbs, 2), "frac_of_pcie": round(per_dev_gbs / pcie, 3),
                     "jpg_bytes": int(sum(sizes[f] for f in range(F))),
                     "lanes": [{"device": s_.device, "frames": s_.frames, "wall_ms": round(s_.wall_ms, 2), "kernel_ms": round(s_.kernel_ms, 2),
                                "staged": s_.staged} for s_ in st],
                     "equal_to_single_frame_entry": bool(ok)}

    F1 = min(frames_per_dev, max_frames)
    Fn = min(frames_per_dev * n_dev, max_frames)
    layouts = []
    if n_dev > 1:
        layouts.append(("devices_1", [0], F1))
    layouts.append(("devices_%d" % n_dev, list(range(n_dev)), Fn))
    if n_dev == 1:
        layouts.append(("two_lanes_on_one_device", [0, 0], F1))
    cache = {}
    for name, devs, F in layouts:
        try:
            if F not in cache:
                cache.clear()
                cache[F] = planes_for(F)
            run(name, devs, F, cache[F])
        except Exception as e:
            res[name] = {"devices": devs, "error": f"{type(e).__name__}: {e}"[:300]}
    # the same with planes the caller has pinned itself: no staging copy, the DMA engines read the caller's memory
    try:
        F = Fn
        if F not in cache:
            cache.clear()
            cache[F] = planes_for(F)
        pinned = [torch.from_numpy(p_).pin_memory() for p_ in cache[F]]
        run("devices_%d_caller_pinned_planes" % n_dev, list(range(n_dev)), F, pinned)
        del pinned
    except Exception as e:
        res["devices_%d_caller_pinned_planes" % n_dev] = {"error": f"{type(e).__name__}: {e}"[:300]}
    # round 5's measurement for comparison: 16 frames, one-shot entry (handle, contexts, rings created and destroyed inside the call)
    try:
        F = 16
        planes = planes_for(F)
        J.encode_batch_multi([0], *planes, W, H, F, chunk_frames=8)
        t0 = time.perf_counter()
        _, jpgs = J.encode_batch_multi([0], *planes, W, H, F, chunk_frames=8)
        dt = time.perf_counter() - t0
        res["one_shot_16_frames"] = {"devices": [0], "frames": F, "ms": round(dt * 1e3, 2), "Mpixels_per_s": round(F * plane / dt / 1e6, 1),
                                     "GBs_h2d": round(3 * plane * F / dt / 1e9, 2), "equal_to_single_frame_entry": bool(jpgs[1] == ref),
                                     "note": "jpezy_encode_batch_multi incl. handle creation + Python-side result slicing, as round 5 timed it"}
    except Exception as e:
        res["one_shot_16_frames"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return res

# ------------------------------------------------------------------------------------------------------------------
def run_rank(args):
    import torch
    import torch.distributed as dist
    import jpezy_amd as J

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the hot path)")
    ndev = torch.cuda.device_count()
    if args.rehearse_on_one_gpu:
        local_rank = 0
    elif local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible "
                         "(--rehearse-on-one-gpu runs the N > 1 control flow on one GPU; its numbers mean nothing)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev                               # where the small reduction tensors live
    multi = world > 1 or args.force_dist       # the distributed calls are made
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        tmo = datetime.timedelta(seconds=args.dist_timeout)      # a collective that hangs raises instead of stalling the job
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", timeout=tmo)
            red_dev = torch.device("cpu")
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)

    W, H, gray, fps, direction, desc = WORKLOADS[args.workload]
    with_entropy = direction == "encode+entropy"
    with_huffdec = direction == "entropy+decode"        # jpezy_read_jpeg_gpu synchronises between its passes: no graph, one stream
    if with_huffdec:
        args.no_graph, args.streams, args.pipelined, args.no_tolerant = True, 1, False, True
    if with_entropy and args.streams > 1:
        # the entropy stage keeps its scratch (tile streams, offsets, unstuffed stream) in the context: two calls in flight on
        # one context would overwrite each other's (include/jpezy_hip.h); frames in flight need a context each -- which is
        # what --pipelined does for this workload (a second context for the odd steps)
        raise SystemExit("--streams is not available for encode4096_jpg: one entropy-stage call in flight per context (use --pipelined)")
    ctx = J.Context(local_rank)
    if args.variant is not None:
        ctx.set_variant(args.variant)
    if args.tolerant:
        ctx.set_decode_tolerance(1)
        args.no_tolerant = True
    ctxs = [ctx]                   # step i runs on ctxs[i % len(ctxs)]; a second context only for the pipelined .jpg measurement
    plane = W * H
    ncoef = J.coeff_count(W, H, gray if direction.startswith("encode") else False)
    step_bytes = algorithmic_bytes(W, H, gray, direction) * fps
    ring = args.ring or max(2, -(-(512 << 20) // step_bytes))

    gen = torch.Generator(device=dev)
    gen.manual_seed(0x6A70657A79 + rank)
    # ring of batches: planar r,g,b [ring][fps][H*W] u8 and coefficients [ring][fps][ncoef] i16
    pr, pg, pb = (torch.randint(0, 256, (ring, fps, plane), dtype=torch.uint8, device=dev, generator=gen) for _ in range(3))
    co = torch.empty((ring, fps, ncoef), dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream(dev)
    jpg = jsz = None
    if with_entropy:
        cap = J.load_library().jpezy_jpeg_bound(W, H)
        jpg = torch.empty((ring, fps, cap), dtype=torch.uint8, device=dev)
        jsz = torch.zeros((ring, fps), dtype=torch.int64, device=dev)

    def enc(i):
        k = i % ring
        ctxs[i % len(ctxs)].fdct_quant_dev(pr[k], pg[k], pb[k], W, H, co[k], gray=gray, n_frames=fps, stream=stream.cuda_stream)

    def entropy(i):
        k = i % ring
        ctxs[i % len(ctxs)].write_jpeg_gpu_dev(co[k], W, H, jpg[k], jsz[k], gray=gray, n_frames=fps, stream=stream.cuda_stream)

    def enc_entropy(i):
        enc(i)
        entropy(i)

    def dec(i):
        k = i % ring
        ctx.dequant_idct_dev(co[k], W, H, pr[k], pg[k], pb[k], gray=gray, n_frames=fps, stream=stream.cuda_stream)

    jfiles = []
    if with_huffdec:
        # real files: encode the ring's random frames to complete .jpg files with the GPU coder, bring the bytes to the host
        # (what decoder::decode reads from disk), then time .jpg -> coefficients -> planes
        cap = J.load_library().jpezy_jpeg_bound(W, H)
        jtmp = torch.empty((fps, cap), dtype=torch.uint8, device=dev)
        jn = torch.zeros((fps,), dtype=torch.int64, device=dev)
        for k in range(ring):
            ctx.fdct_quant_dev(pr[k], pg[k], pb[k], W, H, co[k], gray=False, n_frames=fps, stream=stream.cuda_stream)
            ctx.write_jpeg_gpu_dev(co[k], W, H, jtmp, jn, gray=False, n_frames=fps, stream=stream.cuda_stream)
            torch.cuda.synchronize(dev)
            jfiles.append([jtmp[f, : int(jn[f])].cpu().numpy().copy() for f in range(fps)])
            co[k].zero_()
        del jtmp, jn

    # jpezy_read_jpeg_gpu works on the CONTEXT's stream (and returns when the coefficients are written: it is host-synchronous);
    # dec(i) reads them on torch's stream.  The reuse of ring slot k by a later huffdec is ordered behind dec's read explicitly.
    ctx_stream = torch.cuda.ExternalStream(J.api.load_library().jpezy_ctx_stream(ctx._h), device=dev) if with_huffdec else None
    slot_read = [None] * ring

    def huffdec(i):
        k = i % ring
        if slot_read[k] is not None:
            ctx_stream.wait_event(slot_read[k])
        for f in range(fps):
            ctx.read_jpeg_gpu_into(jfiles[k][f], co[k][f])

    def huffdec_dec(i):
        huffdec(i)
        dec(i)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        slot_read[i % ring] = ev

    if with_huffdec:
        step = huffdec_dec
    elif direction == "decode":
        for k in range(ring):      # real coefficients: encode the random frames once (6-block layout), then time the decode
            ctx.fdct_quant_dev(pr[k], pg[k], pb[k], W, H, co[k], gray=False, n_frames=fps, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        step = dec
    elif with_entropy:
        step = enc_entropy
    else:
        step = enc

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    ctx.fallback_count()           # reset the counter

    def capture(step_fn, streams=1):
        nonlocal stream
        g = torch.cuda.CUDAGraph()
        cap_stream = torch.cuda.Stream(dev)
        cap_stream.wait_stream(stream)
        with torch.cuda.stream(cap_stream):
            stream_saved, stream = stream, cap_stream
            with torch.cuda.graph(g, stream=cap_stream):
                if streams > 1:
                    side = [torch.cuda.Stream(dev) for _ in range(streams)]
                    for sd in side:
                        sd.wait_stream(cap_stream)                  # fork
                    for i in range(args.steps):
                        stream = side[i % streams]
                        step_fn(i)
                    for sd in side:
                        cap_stream.wait_stream(sd)                  # join
                else:
                    for i in range(args.steps):
                        step_fn(i)
            stream = stream_saved
        torch.cuda.synchronize(dev)
        g.replay()                 # instantiate / upload outside the timed region
        torch.cuda.synchronize(dev)
        return g

    graph = None if args.no_graph else capture(step, args.streams)
    ctx.fallback_count()           # reset: only the timed replays are counted

    # ---- the timed region: R replays of exactly K steps, each bracketed by barrier + synchronize on both sides ----
    R = max(1, args.repeats)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(R)]
    wall = []
    for ev0, ev1 in evs:       # every repetition is the contract's sequence: W untimed warm-up steps, then the bracketed K steps
        if args.rewarm:
            for i in range(args.warmup):
                step(i)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev0.record(stream)
        if graph is not None:
            graph.replay()
        else:
            for i in range(args.steps):
                step(i)
        ev1.record(stream)
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        wall.append(time.perf_counter() - t0)
    evms = [ev0.elapsed_time(ev1) for ev0, ev1 in evs]
    nfallback = ctx.fallback_count()

    t = torch.tensor([wall, evms], dtype=torch.float64, device=red_dev)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)       # per replay: the slowest rank
    wall, evms = t[0].tolist(), t[1].tolist()
    elapsed = statistics.median(wall)
    ev_ms = statistics.median(evms)

    # A short K (the driver's --steps 20 is a 0.6 ms region) pays the start of a burst -- launch latency and a cold front end,
    # about 2 us per step at K = 20 -- inside its bracket.  Beside `value` (never instead of it) the same launches are timed
    # over a long replay, so that the line also carries the steady rate the rocprofv3 summaries show.
    long_run = None
    if rank == 0 and graph is not None and args.steps < 100 and not with_entropy and args.streams == 1:
        k_saved, args.steps = args.steps, 200
        gl = capture(step, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lms = []
        for _ in range(5):
            torch.cuda.synchronize(dev)
            e0.record(stream)
            gl.replay()
            e1.record(stream)
            torch.cuda.synchronize(dev)
            lms.append(e0.elapsed_time(e1) / args.steps)
        args.steps = k_saved
        ctx.fallback_count()
        lm = statistics.median(lms)
        long_run = {"steps": 200, "replays": 5, "avg_launch_ms_hip_events": round(lm, 5),
                    "frac": round(step_bytes / (lm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "same launches, one replay of 200 steps (median of 5); this rank only; not part of value"}
        del gl

    # Decode workloads: the opt-in tolerance mode (jpezy_ctx_set_decode_tolerance: FP32 luma without guard band, chroma exact;
    # every byte within one of the reference's, north_star's own bar) timed over the same ring -- an object of its own beside
    # `value`, never instead of it.  max_abs_diff is measured here, against the bit-exact kernel's output of the same frames.
    dec_tol = None
    if rank == 0 and direction == "decode" and graph is not None and args.streams == 1 and not args.no_tolerant:
        dec(0)                                                       # the exact kernel's planes of ring slot 0
        torch.cuda.synchronize(dev)
        exact = [t[0].clone() for t in (pr, pg, pb)]
        ctx.set_decode_tolerance(1)
        k_saved, args.steps = args.steps, 200
        for i in range(args.warmup):
            step(i)
        gt = capture(step, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tms = []
        for _ in range(5):
            torch.cuda.synchronize(dev)
            e0.record(stream)
            gt.replay()
            e1.record(stream)
            torch.cuda.synchronize(dev)
            tms.append(e0.elapsed_time(e1) / args.steps)
        args.steps = k_saved
        diff = max(int((a.to(torch.int16) - t[0].to(torch.int16)).abs().max()) for a, t in zip(exact, (pr, pg, pb)))
        ndiff = sum(int((a != t[0]).sum()) for a, t in zip(exact, (pr, pg, pb)))
        ctx.set_decode_tolerance(0)
        ctx.fallback_count()
        tm = statistics.median(tms)
        dec_tol = {"dtype": "f32 luma + f64 chroma", "max_abs_diff": diff, "tolerance": 1,
                   "bytes_differing": round(ndiff / (3 * plane * fps), 6),
                   "avg_launch_ms_hip_events": round(tm, 5), "value": round(plane * fps / (tm * 1e-3) / 1e6, 2), "unit": "Mpixels/s",
                   "frac": round(step_bytes / (tm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "steps": 200, "replays": 5,
                   "note": "opt-in jpezy_ctx_set_decode_tolerance(ctx, 1): luma IDCT in FP32 without guard band, chroma and colour "
                           "conversion exact; max_abs_diff per channel against the bit-exact kernel on the same frame; this rank "
                           "only; not part of value"}
        del gt, exact

    # besides the vendor peak, report a device-copy bandwidth measured on this box with the same byte count
    # and the same ring (one plain copy kernel per step: half the bytes read, half written)
    copy_gbs = None
    if rank == 0:
        half = step_bytes // 2
        src_bufs = torch.empty((ring, half), dtype=torch.uint8, device=dev)
        dst_bufs = torch.empty((ring, half), dtype=torch.uint8, device=dev)
        for i in range(5):
            dst_bufs[i % ring].copy_(src_bufs[i % ring])
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        c0.record(stream)
        for i in range(reps):
            dst_bufs[i % ring].copy_(src_bufs[i % ring])
        c1.record(stream)
        torch.cuda.synchronize(dev)
        copy_gbs = 2 * half * reps / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del src_bufs, dst_bufs

    huffdec_roof = None
    # Huffman decoder alone (workload decode4096_jpg): K calls of jpezy_read_jpeg_gpu, HIP events on its stream around them.  The
    # call parses the header on the host, uploads the scan and synchronises between its passes, so this is the whole call.
    if with_huffdec and rank == 0:
        # the call is host-synchronous and works on the context's own stream: wall time around a device synchronise is the
        # honest clock (events on torch's stream would bracket nothing of it)
        hms, passes = [], 0
        for _ in range(R):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(args.steps):
                huffdec(i)
            torch.cuda.synchronize(dev)
            hms.append((time.perf_counter() - t0) * 1e3)
            passes = ctx.last_huffdec_passes()
        h_ms = statistics.median(hms) / args.steps
        jbytes = sum(a.size for a in jfiles[0])
        hbytes = jbytes + ncoef * 2 * fps                 # the .jpg read once + the coefficients written once
        ha = hbytes / (h_ms * 1e-3) / 1e9
        huffdec_roof = {"bound": "hbm", "achieved": round(ha, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ha / HBM_PEAK_GBS, 5), "traffic": None,
                        "kernel": "jpezy_read_jpeg_gpu (unstuff + speculate + confirm/refine passes + emit + DC prefix), the whole call",
                        "algorithmic_bytes_per_step": hbytes, "avg_step_ms_wall": round(h_ms, 5),
                        "jpg_bytes_per_frame": jbytes // fps, "sync_passes": passes,
                        "note": "a chain of dependent launches with host decisions between them: latency-bound, not bandwidth-bound; "
                                "the .jpg bytes start on the host (decoder::decode reads a file), the coefficients stay in HBM"}

    # entropy stage alone (workload encode4096_jpg): its own launch sequence, timed with HIP events on the same stream
    entropy_roof = None
    if with_entropy and rank == 0:
        torch.cuda.synchronize(dev)
        sizes = jsz[0].tolist()
        if min(sizes) <= 0:
            raise SystemExit(f"GPU entropy stage reported {sizes}")
        g3 = capture(entropy)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ems = []
        for _ in range(R):
            torch.cuda.synchronize(dev)
            e0.record(stream)
            g3.replay()
            e1.record(stream)
            torch.cuda.synchronize(dev)
            ems.append(e0.elapsed_time(e1))
        ent_ms = statistics.median(ems) / args.steps
        ebytes = ncoef * 2 * fps + int(sum(sizes))         # coefficients read once + the .jpg bytes written once
        ea = ebytes / (ent_ms * 1e-3) / 1e9
        entropy_roof = {"bound": "hbm", "achieved": round(ea, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ea / HBM_PEAK_GBS, 4), "traffic": None,
                        "kernel": "entropy stage (code_tiles + assemble + stuff), all launches of one frame",
                        "algorithmic_bytes_per_step": ebytes, "avg_step_ms_hip_events": round(ent_ms, 5),
                        "jpg_bytes_per_frame": int(sum(sizes) / len(sizes))}
        del g3

    # Pipeline-level number, reported beside `value` and never part of it: the same K steps with two frames in flight
    pipelined = None
    if args.pipelined and rank == 0 and graph is not None and args.streams == 1 and ring >= 2:
        if with_entropy:           # a context of its own for the odd steps (scratch and header cache allocated outside the capture)
            ctx2 = J.Context(local_rank)
            if args.variant is not None:
                ctx2.set_variant(args.variant)
            ctxs.append(ctx2)
            for i in range(4):
                step(i)
            torch.cuda.synchronize(dev)
        g2 = capture(step, 2)
        p0 = time.perf_counter()
        g2.replay()
        torch.cuda.synchronize(dev)
        pdt = time.perf_counter() - p0
        ctx.fallback_count()
        pipelined = {"frames_in_flight": 2, "ms_per_step": round(pdt * 1e3 / args.steps, 5),
                     "value": round(plane * fps * args.steps / pdt / 1e6, 2), "unit": "Mpixels/s",
                     "note": "this rank only; measured after the timed region; not part of value"
                             + ("; two contexts, one per stream" if with_entropy else "")}
        del g2
        if with_entropy:
            torch.cuda.synchronize(dev)
            ctxs.pop().close()

    batch = None
    if (multi and not args.no_batch) or args.batch:
        del graph
        graph = None
        del pr, pg, pb, co
        torch.cuda.empty_cache()
        try:
            batch = run_batch(args, torch, dist, J, ctx, dev, rank, world, world > 1, args.rehearse_on_one_gpu)
        except Exception as e:      # the batch object must never cost the job its headline line
            batch = {"error": f"{type(e).__name__}: {e}"[:400]} if rank == 0 else None
            print(f"bench.py rank {rank}: batch measurement failed: {e}", file=sys.stderr)

    others = None
    if rank == 0 and world == 1 and args.workload == "encode4096" and not args.no_others and args.variant in (None, 1) and not args.tolerant:
        try:
            others = measure_other_workloads(torch, J, ctx, dev, copy_gbs=copy_gbs)
        except Exception as e:
            others = {"error": f"{type(e).__name__}: {e}"[:300]}

    native = None
    if rank == 0 and not args.no_native_multi and args.workload == "encode4096" and not args.rehearse_on_one_gpu:
        try:
            native = measure_native_multi(J, ctx, min(world, torch.cuda.device_count()))
        except Exception as e:
            native = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        px_per_step = plane * fps
        total_px = px_per_step * args.steps * world
        ms_per_step = elapsed * 1e3 / args.steps
        kern_ms = ev_ms / args.steps
        achieved = step_bytes / (kern_ms * 1e-3) / 1e9
        traffic = tsrc = None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                traffic = tj.get(args.workload, {}).get("bytes_per_launch")
                tsrc = tj.get(args.workload, {}).get("source")
            except Exception:
                traffic = None
        if with_entropy or with_huffdec:
            traffic = None
        metric = {"encode": "Mpixels/s encode (FDCT+quant)", "decode": "Mpixels/s decode (dequant+IDCT)",
                  "encode+entropy": "Mpixels/s encode (FDCT+quant + GPU Huffman stage)",
                  "entropy+decode": "Mpixels/s decode (GPU Huffman decoder + dequant+IDCT)"}[direction]
        kernel = {None: "f32::fdct_quant_f32_kernel", 1: "f32::fdct_quant_f32_kernel", 0: "fdct_quant_kernel",
                  2: "f32::fdct_quant_f32_ps_kernel", 3: "f32::fdct_quant_f32_ps2_kernel"}[args.variant] if direction.startswith("encode") else "dequant_idct_kernel"
        out = {
            "metric": metric,
            "value": round(total_px / elapsed / 1e6, 2),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "ms_per_step_min": round(min(wall) * 1e3 / args.steps, 5),
            "ms_per_step_max": round(max(wall) * 1e3 / args.steps, 5),
            "repeats": R,
            "timing": f"median of {R} replays of exactly {args.steps} steps; each replay bracketed by barrier + synchronize, max over ranks",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # the arithmetic the path computes in: encode variant 1 = FP32 first level (FP64 only on guard-band hits),
            # encode variant 0 and decode = FP64; either way the results are the reference's FP64 results bit for bit
            "dtype": ("f32 luma + f64 chroma (tolerance mode, within 1 LSB)" if args.tolerant and direction == "decode" else
                      "f64" if not direction.startswith("encode") or args.variant == 0 else "f32+f64 guard"),
            "data": "synthetic",
            "config": {"workload": desc, "name": args.workload, "width": W, "height": H, "mode": "gray" if gray else "color",
                       "frames_per_step": fps, "ring_batches": ring, "pixels_per_step_per_gpu": px_per_step,
                       "inputs": "iid uniform u8 r,g,b planes resident in HBM", "parallelism": f"frames x{world}",
                       "submission": ("jpezy_read_jpeg_gpu call + IDCT launch per step" if with_huffdec else
                                      "one launch per step" if not with_entropy else "FDCT launch + entropy-stage launches per step")
                                     + ("" if args.no_graph else ", the K steps captured once and replayed as a hipGraph")
                                     + ("" if args.streams <= 1 else f", {args.streams} steps in flight on {args.streams} streams")},
            "exact_fallbacks_per_step": round(nfallback / max(1, args.steps * R), 2),
        }
        if with_huffdec:
            out["roofline"] = huffdec_roof
            out["config"]["inputs"] = ".jpg files of iid uniform u8 frames (GPU coder's output), bytes on the host; coefficients and planes in HBM"
        elif not with_entropy:
            out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                               "kernel": kernel,
                               "algorithmic_bytes_per_launch": step_bytes, "avg_launch_ms_hip_events": round(kern_ms, 5),
                               "avg_launch_ms_min_max": [round(min(evms) / args.steps, 5), round(max(evms) / args.steps, 5)],
                               "device_copy_GBs_measured": round(copy_gbs, 1), "frac_of_device_copy": round(achieved / copy_gbs, 4)}
            if long_run:
                out["roofline"]["long_run"] = long_run
        else:
            out["roofline"] = entropy_roof
            out["roofline"]["note"] = ("the step's dominant cost is the entropy stage; the FDCT kernel's own roofline is the "
                                       "default workload's line")
        if pipelined:
            out["pipelined"] = pipelined
        if dec_tol:
            out["decode_tolerant"] = dec_tol
        if batch:
            out["batch"] = batch
        if batch and "error" not in batch:
            # north_star's 8-GPU target is STRONG scaling of this batch (configs[3]); `value` above is weak scaling of configs[1].
            # Top-level copies so that a 1 -> 8 curve of the target can be read off the lines without opening the object.
            try:
                out["batch_strong_scaling"] = {
                    "metric": "Mpixels/s, batch of %d 1920x1080 frames end to end in .jpg on rank 0 (strong scaling)" % batch["frames"],
                    "value": batch["end_to_end_jpg"]["Mpixels_per_s"],
                    "value_coefficients_gathered": batch["end_to_end"]["Mpixels_per_s"],
                    "value_kernels_only": batch["kernel_only"]["Mpixels_per_s"],
                    "one_gpu_same_pipeline": batch["one_gpu_same_pipeline_jpg"]["Mpixels_per_s"],
                    "speedup_vs_one_gpu": batch["speedup_end_to_end_jpg"],
                    "n_gpus": world}
            except (KeyError, TypeError) as e:      # a partial batch object must not cost the job its headline line either
                out["batch_strong_scaling"] = {"error": f"{type(e).__name__}: {e}"}
        if others:
            # BASELINE configs[2] and [4] (and the decoder's opt-in tolerance mode) as this same run measured them after the timed
            # region: objects of their own, never part of `value`
            out["other_workloads"] = others
        if native:
            out["native_multi_gpu"] = native
        if not args.no_cpu:          # rank 0 only, after every timed region (the other ranks idle at the closing barrier)
            out["cpu_baseline"] = cpu_baseline(W, H, gray, direction)
        print(json.dumps(out), flush=True)

    ctx.close()
    if multi:
        dist.destroy_process_group()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=11, help="replays of the K-step sequence; ms_per_step is their median")
    ap.add_argument("--rewarm", type=int, default=1, help="1: the W warm-up steps are run again before every repetition's bracket "
                    "(each repetition then is `W warm-up steps, K timed steps`, as the first one is); 0: once, before the first")
    ap.add_argument("--workload", default="encode4096", choices=sorted(WORKLOADS))
    ap.add_argument("--ring", type=int, default=0, help="distinct batches in the ring (0 = enough to exceed 512 MiB)")
    ap.add_argument("--variant", type=int, default=None, help="encode kernel variant (0 FP64 butterflies, 1 packed-FP32 first level, one quad per wave [default], 2 / 3 the same arithmetic in persistent workgroups: loader waves + LDS ring / register prefetch)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-native-multi", action="store_true",
                    help="default workload: skip the native_multi_gpu object (jpezy_encode_batch_multi timed by rank 0 after everything else)")
    ap.add_argument("--no-others", action="store_true",
                    help="default workload at N = 1: skip the other_workloads object (configs[2] and [4] measured after the timed region)")
    ap.add_argument("--batch", action="store_true", help="measure the configs[3] batch pipeline also at N = 1")
    ap.add_argument("--no-batch", action="store_true", help="N > 1: skip the configs[3] batch measurement")
    ap.add_argument("--batch-frames", type=int, default=None,
                    help=f"frames of the configs[3] batch (default {BATCH_FRAMES}; 16 in --rehearse-on-one-gpu)")
    ap.add_argument("--batch-chunk", type=int, default=64, help="frames per launch / per transfer of the batch pipeline")
    ap.add_argument("--batch-jpg-stride", type=int, default=2 << 20,
                    help="bytes reserved per frame in the .jpg staging buffers of the batch pipeline (a 1920x1080 random-pixel "
                         "frame codes to ~0.66 MB; a frame that does not fit is reported, never truncated)")
    ap.add_argument("--dist-timeout", type=int, default=300, help="seconds after which a hanging collective raises (N > 1)")
    ap.add_argument("--streams", type=int, default=1,
                    help="frames in flight: step i is launched on stream i %% S (forked from and joined to the timed stream "
                         "inside the captured graph; every step has its own ring buffers).  Default 1: launches back to back")
    ap.add_argument("--pipelined", action="store_true",
                    help="after the timed region, replay the same K steps with two frames in flight and report the result as "
                         "a 'pipelined' object beside value (off by default: the default command launches nothing but the "
                         "timed kernel, so that a rocprofv3 trace of it averages exactly the launches value is made of)")
    ap.add_argument("--no-tolerant", action="store_true",
                    help="decode workloads: skip the decode_tolerant object (the opt-in tolerance mode timed after the timed region)")
    ap.add_argument("--tolerant", action="store_true",
                    help="development aid: run the WHOLE decode bench in the opt-in tolerance mode (value is then not the bit-exact path's)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch the timed steps one by one instead of replaying them as one captured hipGraph")
    ap.add_argument("--force-dist", action="store_true",
                    help="development aid: initialise torch.distributed (nccl = RCCL) even with one rank")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: run the N > 1 control flow with every rank on GPU 0, the gloo backend and host "
                         "staging (RCCL refuses two ranks on one device); the numbers mean nothing")
    args = ap.parse_args(argv)
    if args.batch_frames is None:
        args.batch_frames = 16 if args.rehearse_on_one_gpu else BATCH_FRAMES
    return args


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: be the launcher.  Nothing above has touched the GPU (torch is not even imported yet).
        sys.exit(spawn_ranks(args.gpus, [sys.executable, str(Path(__file__).resolve()), *argv]))
    run_rank(args)


if __name__ == "__main__":
    main()
