"""BASELINE configs[3] on the GPU (run with -m gpu): the batch of 4096 1920x1080 frames at FULL size on one GPU through the
same chunked pipeline bench.py times (jpezy_amd.sharding.gather_to_root_pipelined, single rank), sampled frames
against the oracle, size-independent properties over the whole batch; and the `bench.py --gpus 2` control flow with
both ranks rehearsing on this one GPU.  The loop being sharded: ref encoder/jpezy_encoder.hpp:55-67, once per frame."""
import importlib.util
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_full_4096_frame_1080p_batch_on_one_gpu(oracle):
    import torch
    import jpezy_amd as J
    from jpezy_amd import sharding
    b = _bench()
    W, H, F = b.BATCH_W, b.BATCH_H, b.BATCH_FRAMES
    dev = torch.device("cuda", 0)
    props = torch.cuda.get_device_properties(dev)
    if props.total_memory < 80 << 30:
        # BASELINE configs[3] is sized for the 288 GB of an MI355X: there a short device is a failure, not a skip
        assert "gfx950" not in getattr(props, "gcnArchName", ""), "an MI355X must hold the configs[3] batch (~55 GB)"
        pytest.skip("needs ~55 GB of HBM (not an MI355X)")
    plane = W * H
    cpf = J.coeff_count(W, H, False)
    assert cpf == 120 * 68 * 6 * 64
    ctx = J.Context(0)
    try:
        pr, pg, pb = b.synth_frames(torch, 0, F, plane, dev)
        calls = []

        def make_enc(chunk):
            def enc(lo, hi, dst):
                calls.append((lo, hi))
                ctx.fdct_quant_dev(pr[lo:hi], pg[lo:hi], pb[lo:hi], W, H, dst, gray=False, n_frames=hi - lo,
                                   plane_stride=plane)
            return enc
        out = sharding.gather_to_root_pipelined(make_enc(64), F, cpf, 64, dev)
        torch.cuda.synchronize(dev)
        assert out.shape == (F, cpf) and calls == sharding.chunk_spans(0, F, 64)
        # sampled frames against the oracle: first, last, chunk borders and random ones
        rng = np.random.default_rng(4096)
        sample = sorted({0, F - 1, 63, 64, 2048} | set(int(x) for x in rng.integers(0, F, size=5)))
        assert len(sample) >= 8
        for f in sample:
            r, g, bb = (p[f].cpu().numpy() for p in (pr, pg, pb))
            want = oracle.encode_coeffs(r, g, bb, W, H).reshape(-1)
            got = out[f].cpu().numpy()
            assert np.array_equal(got, want), f"frame {f} of the batch differs from the oracle"
        # a frame is a function of its own pixels only: frame f of the batch generator equals a fresh generation
        r2, _, _ = b.synth_frames(torch, 4095, 4096, plane, dev)
        assert torch.equal(r2[0], pr[F - 1])
        # size-independent properties over the whole batch: (i) the chunking does not matter (one launch of 512 frames
        # = 8 launches of 64); (ii) the shards of an 8-rank split, encoded on their own, tile the batch exactly
        out2 = torch.empty((512, cpf), dtype=torch.int16, device=dev)
        ctx.fdct_quant_dev(pr[1024:1536], pg[1024:1536], pb[1024:1536], W, H, out2, gray=False, n_frames=512, plane_stride=plane)
        torch.cuda.synchronize(dev)
        assert torch.equal(out2, out[1024:1536])
        lo, hi = sharding.shard_range(F, 8, 7)
        assert (lo, hi) == (3584, 4096)
        part = sharding.gather_to_root_pipelined(
            lambda a, c, dst: ctx.fdct_quant_dev(pr[lo + a:lo + c], pg[lo + a:lo + c], pb[lo + a:lo + c], W, H, dst,
                                                 gray=False, n_frames=c - a, plane_stride=plane), hi - lo, cpf, 100, dev)
        torch.cuda.synchronize(dev)
        assert torch.equal(part, out[lo:hi])
        # the padded bottom MCU row (1080 -> 1088 by edge replication, ref :101) is present in every frame
        assert int((out.view(F, 68, 120, 6, 64)[:, 67].abs().sum(dim=(1, 2, 3)) > 0).sum()) == F
    finally:
        ctx.close()


def test_bench_gpus_2_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` starts its own two ranks (no launcher); on this one-GPU box they rehearse on GPU 0
    with gloo + host staging.  Checks the control flow and that the sharded, gathered batch equals the one-GPU batch."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "3",
                        "--warmup", "1", "--repeats", "2", "--no-cpu", "--batch-frames", "10", "--batch-chunk", "3"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["repeats"] == 2
    bt = line["batch"]
    assert bt["frames"] == 10 and bt["frames_per_rank"] == 5 and bt["gathered_equals_one_gpu_result"] is True
    assert bt["kernel_only"]["ms"] > 0 and bt["end_to_end"]["ms"] > 0 and bt["gather"]["bytes_into_rank0"] == 5 * 120 * 68 * 6 * 64 * 2
    assert "rehearsal" in bt
    # the pipeline that moves .jpg files instead of coefficients: same files as the one-GPU run, a tenth of the bytes
    ej = bt["end_to_end_jpg"]
    assert bt["gathered_jpg_equals_one_gpu_result"] is True and ej["ms"] > 0 and bt["one_gpu_same_pipeline_jpg"]["ms"] > 0
    assert 0 < ej["bytes_into_rank0"] < bt["gather"]["bytes_into_rank0"] // 5 and ej["bytes_into_rank0"] < ej["jpg_bytes"]


def test_jpg_batch_pipeline_on_one_gpu(oracle):
    """gather_jpg_to_root_pipelined, single rank, on the device: FDCT + GPU Huffman stage chunk by chunk through a staging
    ring shorter than the number of chunks; every file equals the oracle's (encoder::encode, ref
    encoder/jpezy_encoder.hpp:38-77); a stride that is too small is an error, not a truncated file."""
    import torch
    import jpezy_amd as J
    from jpezy_amd import sharding
    b = _bench()
    W, H, F, chunk = 208, 120, 11, 3
    dev = torch.device("cuda", 0)
    plane, cpf = W * H, J.coeff_count(W, H, False)
    ctx = J.Context(0)
    try:
        pr, pg, pb = b.synth_frames(torch, 0, F, plane, dev)
        stride = 64 << 10
        jbuf = [torch.empty((chunk, stride), dtype=torch.uint8, device=dev) for _ in range(2)]
        jsz = [torch.zeros(chunk, dtype=torch.int64, device=dev) for _ in range(2)]
        jco = torch.empty((chunk, cpf), dtype=torch.int16, device=dev)

        def enc(lo, hi, slot, bufs=jbuf):
            n = hi - lo
            ctx.fdct_quant_dev(pr[lo:hi], pg[lo:hi], pb[lo:hi], W, H, jco[:n], n_frames=n, plane_stride=plane)
            ctx.write_jpeg_gpu_dev(jco[:n], W, H, bufs[slot][:n], jsz[slot][:n], n_frames=n)
            return bufs[slot][:n], jsz[slot][:n]
        res = sharding.gather_jpg_to_root_pipelined(enc, F, chunk, dev, ring=2)
        torch.cuda.synchronize(dev)
        assert [(c[0], c[1]) for c in res.chunks] == sharding.chunk_spans(0, F, chunk)
        for f in range(F):
            r, g, bb = (p[f].cpu().numpy() for p in (pr, pg, pb))
            assert res.frame(f).cpu().numpy().tobytes() == oracle.encode_jpeg(r, g, bb, W, H, False), f
        small = [torch.empty((chunk, 1024), dtype=torch.uint8, device=dev) for _ in range(2)]
        with pytest.raises(RuntimeError):
            sharding.gather_jpg_to_root_pipelined(lambda lo, hi, slot: enc(lo, hi, slot, small), F, chunk, dev, ring=2)
    finally:
        ctx.close()


def test_bench_gpus_2_on_rccl_when_two_gpus_are_visible():
    """the first multi-GPU box exercises RCCL inside the suite: `bench.py --gpus 2` on the nccl backend (one rank per GPU),
    a small batch; both gathers must reproduce the one-GPU results.  Skipped on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: RCCL needs a device per rank (bench.py --rehearse-on-one-gpu covers the control flow)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--repeats", "3",
                        "--no-cpu", "--batch-frames", "96", "--batch-chunk", "16"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    bt = line["batch"]
    assert line["n_gpus"] == 2 and "error" not in bt and "rehearsal" not in bt
    assert bt["gathered_equals_one_gpu_result"] is True and bt["gathered_jpg_equals_one_gpu_result"] is True
    assert bt["end_to_end_jpg"]["bytes_into_rank0"] < bt["gather"]["bytes_into_rank0"] // 5


def test_native_entry_between_two_different_gpus_when_two_are_visible(oracle):
    """the native entry's peer-copy branch with src != dst (jpezy_capi_multi.hip: hipMemcpyPeerAsync from a non-root lane's slot into
    devices[0]'s memory) and the host delivery of two lanes over two PCIe links: results equal to the oracle's, lane statistics name two
    devices.  No one-GPU box can run this: skipped there -- the first multi-GPU box that runs the suite is its first execution."""
    import torch
    import jpezy_amd as J
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: traffic between two different GPUs needs two (tests/test_gpu_multi.py runs the lanes on one device)")
    W, H, F = 256, 80, 13
    frames = [oracle.synth_rgb(W, H, frame=7000 + f) for f in range(F)]
    planes = [np.concatenate([fr[k] for fr in frames]) for k in range(3)]
    want = np.stack([oracle.encode_coeffs(*fr, W, H) for fr in frames])
    with J.MultiEncoder([0, 1], W, H, chunk_frames=2) as M:
        for on_root in (True, False):
            co, jpg = M.encode(*planes, F, want_coeffs=True, on_root_device=on_root)
            assert np.array_equal(co.reshape(want.shape), want), on_root
            for f in range(F):
                assert jpg[f] == oracle.write_jpeg(want[f], W, H, False), (f, on_root)
            st = M.stats()
            assert [s_["device"] for s_ in st] == [0, 1] and [s_["frames"] for s_ in st] == [7, 6]
    with J.MultiEncoder([1, 0], W, H) as M:                                # the root need not be device 0
        co, jpg = M.encode(*planes, F, want_coeffs=True, on_root_device=True)
        assert np.array_equal(co.reshape(want.shape), want)


@pytest.mark.parametrize("W,H,gray,band_rows", [(4096, 4096, False, 32), (7680, 4320, True, 17), (1000, 530, False, 5)])
def test_one_frame_split_by_mcu_row_bands_on_the_gpu(oracle, W, H, gray, band_rows):
    """configs[1] / [4] as ONE frame cut into MCU-row bands (jpezy_amd.sharding.encode_frame_banded, the N > 1 form of a single
    frame; here one rank walks all bands): every band is a kernel launch on a slice of the planes, the assembled coefficient
    buffer equals the whole-frame launch bit for bit -- also for the last, shorter band of 4320 = 270 MCU rows and for a width
    that takes the unaligned kernel -- and a sample of MCU rows equals the oracle"""
    import torch
    import jpezy_amd as J
    from jpezy_amd import sharding
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(W * 7 + H)
    planes = [torch.randint(0, 256, (W * H,), dtype=torch.uint8, device=dev, generator=g) for _ in range(3)]
    ctx = J.Context(0)
    try:
        whole = torch.empty(J.coeff_count(W, H, gray), dtype=torch.int16, device=dev)
        ctx.fdct_quant_dev(planes[0], planes[1], planes[2], W, H, whole, gray=gray)
        bands = []

        def encode_band(lo, hi, dst):
            y0, n = sharding.band_pixel_rows(lo, hi, H)
            bands.append((lo, hi))
            sl = slice(y0 * W, (y0 + n) * W)
            ctx.fdct_quant_dev(planes[0][sl], planes[1][sl], planes[2][sl], W, n, dst.reshape(-1), gray=gray)

        got = sharding.encode_frame_banded(encode_band, W, H, gray=gray, band_rows=band_rows, device=dev)
        torch.cuda.synchronize(dev)
        mcu_rows = (H + 15) // 16
        assert bands == sharding.chunk_spans(0, mcu_rows, band_rows)
        assert torch.equal(got.reshape(-1), whole)
        # two bands against the oracle, as frames of their own (first and last)
        hp = [p.cpu().numpy() for p in planes]
        for lo, hi in (bands[0], bands[-1]):
            y0, n = sharding.band_pixel_rows(lo, hi, H)
            sl = slice(y0 * W, (y0 + n) * W)
            want = oracle.encode_coeffs(hp[0][sl], hp[1][sl], hp[2][sl], W, n, gray=gray)
            assert np.array_equal(got[lo:hi].cpu().numpy().reshape(-1), np.ascontiguousarray(want).reshape(-1))
    finally:
        ctx.close()


_RCCL_WORLD1 = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import jpezy_amd as J
from jpezy_amd import sharding
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=0, world_size=1)
try:
    W, H, F = 256, 64, 5
    cpf = J.coeff_count(W, H)
    g = torch.Generator(device=dev); g.manual_seed(7)
    planes = [torch.randint(0, 256, (F, W * H), dtype=torch.uint8, device=dev, generator=g) for _ in range(3)]
    ctx = J.Context(0)
    local = torch.empty((F, cpf), dtype=torch.int16, device=dev)
    ctx.fdct_quant_dev(planes[0], planes[1], planes[2], W, H, local, n_frames=F, plane_stride=W * H)
    # the monolithic gather is ONE RCCL all_gather (world 1: a device-to-device copy by an RCCL kernel of this process's
    # communicator); padded shards as at N > 1
    full = sharding.gather_coefficients(local.reshape(-1), F, cpf)
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    assert torch.equal(full, local) and float(t[0]) == 1.0
    print("rccl world 1 ok", dist.get_backend())
finally:
    dist.destroy_process_group()
'''


def test_rccl_communicator_and_all_gather_on_one_gpu(tmp_path):
    """What a one-GPU box can execute of RCCL: the nccl backend's communicator is created for this GPU and the monolithic
    coefficient gather (`sharding.gather_coefficients`: one all_gather, byte-typed, padded shards) and an all_reduce run through
    it at world size 1 -- the library, the HSA IPC mode and torch's binding are exercised; links between GPUs are not (no
    multi-GPU number exists).  In a subprocess: the default process group must not leak into the suite."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "rccl_world1.py"
    script.write_text(_RCCL_WORLD1)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, str(script), str(ROOT), str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "rccl world 1 ok nccl" in p.stdout
