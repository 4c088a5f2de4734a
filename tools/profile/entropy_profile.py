import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch, ctypes as C
import jpezy_amd as J
from jpezy_amd import api
ctx = J.Context(0); lib = api.load_library()
W = H = 4096; n = 1
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(7)
planes = [torch.randint(0, 256, (n, W * H), dtype=torch.uint8, device=dev, generator=g) for _ in range(3)]
co = torch.empty((n, J.coeff_count(W, H, False)), dtype=torch.int16, device=dev)
ctx.fdct_quant_dev(planes[0], planes[1], planes[2], W, H, co, n_frames=n); torch.cuda.synchronize()
cap = 8 << 20
buf = np.zeros(cap * n, dtype=np.uint8); sizes = (C.c_long * n)()
for _ in range(6):
    assert lib.jpezy_write_jpeg_gpu_batch(ctx._h, co.data_ptr(), W, H, 0, n, b"Encoded by jpezy", buf.ctypes.data, cap, sizes) == 0
print(sizes[0])
