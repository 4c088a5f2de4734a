"""jpezy_amd -- MI355X-native (gfx950) baseline-JPEG hot path of falgon/jpezy.

The product is the C-ABI shared library `libjpezy_hip.so` (include/jpezy_hip.h): hand-written HIP kernels
for RGB->YCbCr + 4:2:0 + FDCT + quantise + zig-zag and dequantise + IDCT + upsample + YCbCr->RGB, plus the
host-side Huffman/JFIF tail.  This package is the Python binding used by tests/ and bench.py (torch is
only plumbing: device memory and streams) and a mirror of the reference's `encoder` / `decoder` surface.

There is no CPU fallback: importing works anywhere, but every compute call needs the built library and
a HIP device and raises JpezyError otherwise.
"""
from .api import (  # noqa: F401
    Context,
    Decoder,
    Encoder,
    FrameInfo,
    JpezyError,
    MultiEncoder,
    coeff_count,
    encode_batch_multi,
    library_path,
    load_library,
    mcu_grid,
    read_jpeg,
    shard_range,
    write_jpeg,
    write_jpeg_batch,
)

__all__ = [
    "Context", "Decoder", "Encoder", "FrameInfo", "JpezyError", "MultiEncoder", "coeff_count", "encode_batch_multi", "library_path",
    "load_library", "mcu_grid", "read_jpeg", "shard_range", "write_jpeg", "write_jpeg_batch",
]
