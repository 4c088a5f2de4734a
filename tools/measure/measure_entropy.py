#!/usr/bin/env python3
"""Development probe: the encoder's tail on the GPU vs on the host (4096x4096 random-pixel frame, 32 x 1080p batch)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402


def main():
    ctx = J.Context(0)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(7)
    for (W, H, n) in ((4096, 4096, 1), (1920, 1080, 32)):
        planes = [torch.randint(0, 256, (n, W * H), dtype=torch.uint8, device=dev, generator=g) for _ in range(3)]
        co = torch.empty((n, J.coeff_count(W, H, False)), dtype=torch.int16, device=dev)
        ctx.fdct_quant_dev(planes[0], planes[1], planes[2], W, H, co, n_frames=n)
        torch.cuda.synchronize()
        out = ctx.write_jpeg_gpu(co, W, H, n_frames=n)
        # C-ABI call with a preallocated, already touched output buffer (the Python wrapper's allocation and
        # bytes() copies are not part of the path)
        import ctypes as C
        from jpezy_amd import api
        lib = api.load_library()
        cap = max(len(o) for o in out) + 1024
        buf = np.zeros(cap * n, dtype=np.uint8)
        sizes = (C.c_long * n)()
        comment = b"Encoded by jpezy"
        def call():
            rc = lib.jpezy_write_jpeg_gpu_batch(ctx._h, co.data_ptr(), W, H, 0, n, comment, buf.ctypes.data, cap, sizes)
            assert rc == 0
        call()
        t = time.perf_counter()
        reps = 10
        for _ in range(reps):
            call()
        dt = (time.perf_counter() - t) / reps
        assert all(bytes(buf[f * cap: f * cap + sizes[f]]) == out[f] for f in range(n))
        # device-resident pipeline: planes in HBM -> whole .jpg files in HBM, FDCT + entropy stage, replayed as a hipGraph
        stride = max(len(o) for o in out) + 4096
        d_out = torch.empty((n, stride), dtype=torch.uint8, device=dev)
        d_sizes = torch.zeros(n, dtype=torch.int64, device=dev)
        ctx.write_jpeg_gpu_dev(co, W, H, d_out, d_sizes, n_frames=n)     # header upload + scratch allocation outside the capture
        torch.cuda.synchronize()
        K = 20
        gph = torch.cuda.CUDAGraph()
        cs = torch.cuda.Stream()
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            with torch.cuda.graph(gph, stream=cs):
                for _ in range(K):
                    ctx.fdct_quant_dev(planes[0], planes[1], planes[2], W, H, co, n_frames=n, stream=cs.cuda_stream)
                    ctx.write_jpeg_gpu_dev(co, W, H, d_out, d_sizes, n_frames=n, stream=cs.cuda_stream)
        gph.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gph.replay(); e1.record(); torch.cuda.synchronize()
        dt_dev = e0.elapsed_time(e1) * 1e-3 / K
        assert all(d_out[f, :int(d_sizes[f])].cpu().numpy().tobytes() == out[f] for f in range(n))
        print(f"{n} x {W}x{H}: planes in HBM -> .jpg files in HBM (FDCT + Huffman + stuffing, all on the device, graph replay): "
              f"{dt_dev * 1e6:.0f} us = {W * H * n / dt_dev / 1e6:.0f} Mpx/s")
        px = W * H * n
        nbytes = sum(len(o) for o in out)
        host = co.cpu().numpy()
        t = time.perf_counter()
        ref = J.write_jpeg_batch(host, W, H, n, threads=1)
        dth = time.perf_counter() - t
        t = time.perf_counter()
        ref16 = J.write_jpeg_batch(host, W, H, n, threads=0)
        dtm = time.perf_counter() - t
        ok = all(a == b for a, b in zip(out, ref))
        print(f"{n} x {W}x{H}: {nbytes / 1e6:.1f} MB of .jpg; GPU tail (device coefficients -> host bytes) {dt * 1e3:.2f} ms = {px / dt / 1e6:.0f} Mpx/s; "
              f"host tail 1 thread {dth * 1e3:.1f} ms = {px / dth / 1e6:.0f} Mpx/s, all cores {dtm * 1e3:.1f} ms; identical bytes: {ok}")


if __name__ == "__main__":
    main()
