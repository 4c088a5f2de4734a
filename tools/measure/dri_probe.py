#!/usr/bin/env python3
"""Development probe: phases of the GPU Huffman decoder on a file with many small restart intervals (JPEZY_BATCH_DEBUG=1)."""
import io
import sys
import time
from pathlib import Path

import numpy as np
from PIL import Image, ImageFile

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import jpezy_amd as J  # noqa: E402

ImageFile.MAXBLOCK = 1 << 26
rng = np.random.default_rng(5)
W = H = 4096
yy, xx = np.mgrid[0:H, 0:W]
img = np.clip((np.sin(xx / 37.0) * 60 + np.cos(yy / 23.0) * 50 + 128)[..., None] + rng.normal(0, 12, (H, W, 3)), 0, 255).astype(np.uint8)
ctx = J.Context(0)
for blocks in (int(a) for a in (sys.argv[1:] or ["8"])):
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", subsampling=0, quality=90, restart_marker_blocks=blocks)
    data = buf.getvalue()
    ctx.read_jpeg_gpu(data)
    t = time.perf_counter()
    ctx.read_jpeg_gpu(data)
    print(f"restart interval {blocks} MCUs: {len(data) / 1e6:.1f} MB, {(time.perf_counter() - t) * 1e3:.2f} ms, {'GPU' if ctx.last_huffdec_passes() else 'HOST'}", file=sys.stderr)
