#!/usr/bin/env python3
"""Development probe: 4096x4096 encode time with every quad forced through one of the exact paths (force_exact 1, 2, 3);
3 = the per-lane evaluator of the queue-overflow case, i.e. the worst case an adversarial image can cause."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
import jpezy_amd as J
ctx = J.Context(0)
W = H = 4096
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
planes = [torch.randint(0, 256, (W * H,), dtype=torch.uint8, device=dev, generator=g) for _ in range(3)]
co = torch.empty(J.coeff_count(W, H, False), dtype=torch.int16, device=dev)
for force in (0, 3, 2, 1):
    ctx.set_force_exact(force)
    ctx.fdct_quant_dev(planes[0], planes[1], planes[2], W, H, co); torch.cuda.synchronize()
    t = time.perf_counter()
    n = 3 if force in (1, 2) else 10
    for _ in range(n):
        ctx.fdct_quant_dev(planes[0], planes[1], planes[2], W, H, co)
    torch.cuda.synchronize()
    print(f"force_exact={force}: {(time.perf_counter() - t) / n * 1e6:.0f} us per 4096^2 frame")
