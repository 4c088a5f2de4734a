#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the fused encode (FDCT+quant) stage on MI355X, with its HBM roofline
fraction and the CPU oracle timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload encode4096|decode4096|gray8k|batch1080p]

A *step* is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
the default workload is BASELINE.json configs[1], one 4096x4096 random-pixel frame per step.  Steps walk
a ring of distinct frames whose inputs+outputs (8 x 100.7 MB) exceed the 256 MiB Infinity Cache, so the
kernel really streams from HBM.  N > 1 (launched by torch.distributed.run, one rank per GPU): frames are
independent, so every rank encodes its own frames (weak scaling, no data-path collective); the RCCL
gather of coefficient buffers that BASELINE.json's north_star asks for is measured separately and
reported under "gather", never inside `value`.  The K timed steps are a fixed sequence of K launches (one per
step, on torch's current stream): they are captured once into a hipGraph and replayed inside the timed region
(--no-graph launches them one by one: same kernels and work, ~1.5 us more launch gap per step).

Prints ONE JSON line (rank 0).  `oracle/` is used only for the cpu_baseline leg.
"""
import argparse
import json
import os
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
}


def algorithmic_bytes(W, H, gray, direction):
    """encode colour 3 B/px read + 1.5 samples x 2 B written; gray 3 + 2; decode colour
    3 B/px of coefficients read + 3 B/px written.  Padded MCU grid for the coefficient side."""
    mc, mr = (W + 15) // 16, (H + 15) // 16
    px = W * H
    if direction == "encode":
        return 3 * px + mc * mr * (4 if gray else 6) * 128
    return mc * mr * 6 * 128 + 3 * px


def cpu_baseline(W, H, gray, budget_s=12.0):
    """The oracle (a scalar C port of the reference's algorithm) on ONE host core, on a bounded band of
    MCU rows of the same workload."""
    import numpy as np
    from oracle import oracle as O
    r, g, b = O.synth_rgb(W, min(H, 512))
    reps = -(-H // min(H, 512))
    r, g, b = (np.tile(p, reps)[: W * H] for p in (r, g, b))
    mr = (H + 15) // 16
    # calibrate on 2 MCU rows, then size the band to the budget
    t0 = time.perf_counter()
    O.encode_coeffs(r, g, b, W, H, gray, rows=(0, 2))
    per_row = (time.perf_counter() - t0) / 2
    rows = max(2, min(mr, int(budget_s / max(per_row, 1e-9))))
    passes = max(1, int(budget_s / max(per_row * rows, 1e-9)))
    t0 = time.perf_counter()
    for _ in range(passes):
        O.encode_coeffs(r, g, b, W, H, gray, rows=(0, rows))
    dt = time.perf_counter() - t0
    px = rows * 16 * W * passes
    return {
        "value": round(px / dt / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
        "sample": f"{passes} pass(es) over {rows} of {mr} MCU rows ({px / 1e6:.2f} Mpx) of the {W}x{H} frame, "
                  f"colour+FDCT+quant+zig-zag stage only, oracle/jpezy_oracle.c -O2 -ffp-contract=off, {dt:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="encode4096", choices=sorted(WORKLOADS))
    ap.add_argument("--ring", type=int, default=0, help="distinct batches in the ring (0 = enough to exceed 512 MiB)")
    ap.add_argument("--variant", type=int, default=None, help="encode kernel variant (0 FP64 butterflies, 1 FP32 first level)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-gather", action="store_true", help="skip the separate RCCL gather measurement (N>1)")
    ap.add_argument("--streams", type=int, default=1,
                    help="frames in flight: step i is launched on stream i %% S (forked from and joined to the timed stream "
                         "inside the captured graph; every step has its own ring buffers).  Default 1: launches back to back; "
                         "with 2 the ramp and drain of consecutive launches overlap (pipeline-level number, see DESIGN.md)")
    ap.add_argument("--pipelined", action="store_true",
                    help="after the timed region, replay the same K steps with two frames in flight and report the result as "
                         "a 'pipelined' object beside value (off by default: the default command launches nothing but the "
                         "timed kernel, so that a rocprofv3 trace of it averages exactly the launches value is made of)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch the timed steps one by one instead of replaying them as one captured hipGraph "
                         "(the graph saves ~1.5 us of launch gap per 30 us step; same kernels, same work)")
    ap.add_argument("--force-dist", action="store_true",
                    help="development aid: initialise torch.distributed (nccl = RCCL) even with one rank, so that the "
                         "barrier / all_reduce / all_gather calls of the N>1 path run on a single-GPU box")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: run the N>1 control flow with every rank on GPU 0 and the gloo backend "
                         "(RCCL refuses two ranks on one device); the numbers mean nothing")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import jpezy_amd as J

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the hot path)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev                       # where the small reduction tensors live
    multi = world > 1 or args.force_dist       # the distributed calls are made
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
            red_dev = torch.device("cpu")
            args.no_gather = True
        else:
            dist.init_process_group("nccl", device_id=dev)

    W, H, gray, fps, direction, desc = WORKLOADS[args.workload]
    ctx = J.Context(local_rank)
    if args.variant is not None:
        ctx.set_variant(args.variant)
    plane = W * H
    ncoef = J.coeff_count(W, H, gray if direction == "encode" else False)
    step_bytes = algorithmic_bytes(W, H, gray, direction) * fps
    ring = args.ring or max(2, -(-(512 << 20) // step_bytes))

    gen = torch.Generator(device=dev)
    gen.manual_seed(0x6A70657A79 + rank)
    # ring of batches: planar r,g,b [ring][fps][H*W] u8 and coefficients [ring][fps][ncoef] i16
    pr, pg, pb = (torch.randint(0, 256, (ring, fps, plane), dtype=torch.uint8, device=dev, generator=gen) for _ in range(3))
    co = torch.empty((ring, fps, ncoef), dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream(dev)

    def enc(i):
        k = i % ring
        ctx.fdct_quant_dev(pr[k], pg[k], pb[k], W, H, co[k], gray=gray, n_frames=fps, stream=stream.cuda_stream)

    def dec(i):
        k = i % ring
        ctx.dequant_idct_dev(co[k], W, H, pr[k], pg[k], pb[k], gray=gray, n_frames=fps, stream=stream.cuda_stream)

    if direction == "decode":
        for k in range(ring):      # real coefficients: encode the random frames once (6-block layout), then time the decode
            ctx.fdct_quant_dev(pr[k], pg[k], pb[k], W, H, co[k], gray=False, n_frames=fps, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        step = dec
    else:
        step = enc

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    ctx.fallback_count()           # reset the counter

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    graph = None
    if not args.no_graph:
        # the K timed steps are a fixed sequence of launches: capture them once, replay them inside the timed region
        graph = torch.cuda.CUDAGraph()
        cap_stream = torch.cuda.Stream(dev)
        cap_stream.wait_stream(stream)
        with torch.cuda.stream(cap_stream):
            stream_saved, stream = stream, cap_stream
            with torch.cuda.graph(graph, stream=cap_stream):
                if args.streams > 1:
                    side = [torch.cuda.Stream(dev) for _ in range(args.streams)]
                    for sd in side:
                        sd.wait_stream(cap_stream)                  # fork
                    for i in range(args.steps):
                        stream = side[i % args.streams]
                        step(i)
                    for sd in side:
                        cap_stream.wait_stream(sd)                  # join
                else:
                    for i in range(args.steps):
                        step(i)
            stream = stream_saved
        torch.cuda.synchronize(dev)
        graph.replay()
        torch.cuda.synchronize(dev)
        ctx.fallback_count()       # reset: only the timed replay is counted
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
    elapsed = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    nfallback = ctx.fallback_count()

    t = torch.tensor([elapsed, ev_ms], dtype=torch.float64, device=red_dev)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, ev_ms = float(t[0]), float(t[1])

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

    # Pipeline-level number, reported beside `value` and never part of it: the same K steps with two frames in flight
    # (two streams inside one graph; every step has its own ring buffers).  The ramp and drain of consecutive launches
    # overlap, which a single in-order stream cannot do.
    pipelined = None
    if args.pipelined and rank == 0 and graph is not None and args.streams == 1 and ring >= 2:
        g2 = torch.cuda.CUDAGraph()
        cap_stream = torch.cuda.Stream(dev)
        cap_stream.wait_stream(stream)
        with torch.cuda.stream(cap_stream):
            stream_saved = stream
            with torch.cuda.graph(g2, stream=cap_stream):
                side = [torch.cuda.Stream(dev) for _ in range(2)]
                for sd in side:
                    sd.wait_stream(cap_stream)
                for i in range(args.steps):
                    stream = side[i % 2]
                    step(i)
                for sd in side:
                    cap_stream.wait_stream(sd)
            stream = stream_saved
        torch.cuda.synchronize(dev)
        g2.replay()
        torch.cuda.synchronize(dev)
        p0 = time.perf_counter()
        g2.replay()
        torch.cuda.synchronize(dev)
        pdt = time.perf_counter() - p0
        ctx.fallback_count()
        pipelined = {"frames_in_flight": 2, "ms_per_step": round(pdt * 1e3 / args.steps, 5),
                     "value": round(plane * fps * args.steps / pdt / 1e6, 2), "unit": "Mpixels/s",
                     "note": "this rank only; measured after the timed region; not part of value"}
        del g2

    gather = None
    if multi and not args.no_gather:
        # north_star: gather the coefficient buffers over xGMI -- jpezy_amd.sharding.gather_coefficients,
        # one all_gather of this rank's step output (fps frames per rank).
        from jpezy_amd import sharding
        src = co[0].reshape(-1)
        for _ in range(3):
            sharding.gather_coefficients(src, fps * world, ncoef)
        torch.cuda.synchronize(dev)
        dist.barrier()
        g0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            sharding.gather_coefficients(src, fps * world, ncoef)
        torch.cuda.synchronize(dev)
        dist.barrier()
        gdt = (time.perf_counter() - g0) / reps
        gt = torch.tensor([gdt], dtype=torch.float64, device=dev)
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        gdt = float(gt[0])
        gather = {"op": "all_gather_into_tensor(int16 coefficients of one step)", "bytes_per_rank": src.numel() * 2,
                  "ms": round(gdt * 1e3, 4), "algbw_GBs": round(src.numel() * 2 * world / gdt / 1e9, 2),
                  "note": "measured after the timed region; not part of value"}

    if rank == 0:
        px_per_step = plane * fps
        total_px = px_per_step * args.steps * world
        ms_per_step = elapsed * 1e3 / args.steps
        kern_ms = ev_ms / args.steps
        achieved = step_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists():
            try:
                traffic = json.loads(tfile.read_text()).get(args.workload)
            except Exception:
                traffic = None
        out = {
            "metric": "Mpixels/s encode (FDCT+quant)" if direction == "encode" else "Mpixels/s decode (dequant+IDCT)",
            "value": round(total_px / elapsed / 1e6, 2),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # the arithmetic the path computes in: encode variant 1 = FP32 first level (FP64 only on guard-band hits),
            # encode variant 0 and decode = FP64; either way the results are the reference's FP64 results bit for bit
            "dtype": "f32+f64 guard" if (direction == "encode" and args.variant in (None, 1)) else "f64",
            "data": "synthetic",
            "config": {"workload": desc, "name": args.workload, "width": W, "height": H, "mode": "gray" if gray else "color",
                       "frames_per_step": fps, "ring_batches": ring, "pixels_per_step_per_gpu": px_per_step,
                       "inputs": "iid uniform u8 r,g,b planes resident in HBM", "parallelism": f"frames x{world}",
                       "submission": "one launch per step" + ("" if args.no_graph else ", the K steps captured once and replayed as a hipGraph")
                                     + ("" if args.streams <= 1 else f", {args.streams} steps in flight on {args.streams} streams")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": ("f32::fdct_quant_f32_kernel" if args.variant in (None, 1) else "fdct_quant_kernel")
                         if direction == "encode" else "dequant_idct_kernel",
                         "algorithmic_bytes_per_launch": step_bytes, "avg_launch_ms_hip_events": round(kern_ms, 5),
                         "device_copy_GBs_measured": round(copy_gbs, 1), "frac_of_device_copy": round(achieved / copy_gbs, 4)},
            "exact_fallbacks_per_step": round(nfallback / max(1, args.steps), 2),
        }
        if pipelined:
            out["pipelined"] = pipelined
        if gather:
            out["gather"] = gather
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(W, H, gray)
        print(json.dumps(out), flush=True)

    ctx.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
