import io, sys, time, numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J
from PIL import Image, ImageFile
ImageFile.MAXBLOCK = 1 << 26
ctx = J.Context(0); ctx.set_huffdec_min_bytes(0)
rng = np.random.default_rng(1)
def timeit(name, data, W, H):
    arr = np.frombuffer(data, dtype=np.uint8).copy()
    co = torch.empty(W * H * 4, dtype=torch.int16, device="cuda:0")
    ctx.read_jpeg_gpu_into(arr, co); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.read_jpeg_gpu_into(arr, co); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); J.read_jpeg(data); th = time.perf_counter() - t0
    print(f"{name}: {len(data)/1024:.0f} KiB, GPU path {np.median(ts)*1e3:.2f} ms ({ctx.last_huffdec_passes()} launches; 0 = host decoder), host decoder alone {th*1e3:.2f} ms")
W = H = 4096
flat = np.full(W * H, 128, np.uint8)
timeit("flat gray 4096^2", ctx.encode_jpeg(flat, flat, flat, W, H), W, H)
W2, H2 = 4080, 4096       # MCU rows of 255 MCUs: the period of the flat stream no longer divides a subsequence
flat2 = np.full(W2 * H2, 77, np.uint8)
timeit("flat 4080x4096 (level 77)", ctx.encode_jpeg(flat2, np.full(W2 * H2, 200, np.uint8), flat2, W2, H2), W2, H2)
half = flat.copy().reshape(H, W); half[: H // 2] = rng.integers(0, 256, (H // 2, W), dtype=np.uint8); half = half.reshape(-1)
timeit("half noise, half flat", ctx.encode_jpeg(half, half, half, W, H), W, H)
yy, xx = np.mgrid[0:H, 0:W]
grad = ((xx + yy) // 32 % 256).astype(np.uint8).reshape(-1)
timeit("slow gradient", ctx.encode_jpeg(grad, grad, grad, W, H), W, H)
img = rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
buf = io.BytesIO(); Image.fromarray(img).save(buf, "JPEG", subsampling=0, quality=95, optimize=True)
timeit("1024^2 noise 4:4:4 q95 optimised", buf.getvalue(), 1024, 1024)
