"""The end of an entropy-coded segment (jpezy_host::entropy_segment_length, the 16-bytes-at-a-time pass that jpezy_read_jpeg_gpu and
jpezy_decode_jpeg_batch make over a scan before it goes to the GPU Huffman decoder) against the plain rule: the segment ends before the
first 0xFF that is followed by anything but 0x00, or that is the last byte (ref decoder/jpezy_decoder.hpp:583-642: the bit reader
swallows 0xFF00 and stops at a marker).  CPU only: the function is host code inside the C-ABI library."""
import ctypes as C

import numpy as np

from jpezy_amd import api


def naive(b):
    n = len(b)
    for i in range(n):
        if b[i] == 0xFF and (i + 1 >= n or b[i + 1] != 0x00):
            return i
    return n


def test_entropy_segment_length_matches_the_plain_rule():
    lib = C.CDLL(str(api.library_path()))
    fn = getattr(lib, "_ZN10jpezy_host22entropy_segment_lengthEPKhm")
    fn.restype, fn.argtypes = C.c_size_t, [C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(11)
    cases = [b"", b"\xff", b"\x00", b"\xff\x00", b"\xff\xd9", b"\x12" * 40 + b"\xff", b"\xff\x00" * 30 + b"\xff\xd9"]
    for n in list(range(0, 70)) + [255, 256, 257, 1000, 4099]:
        for _ in range(6):
            a = rng.integers(0, 256, n, dtype=np.uint8)
            a[rng.random(n) < 0.2] = 0xFF                       # many candidates ...
            nxt = np.flatnonzero(a[:-1] == 0xFF) + 1 if n > 1 else np.array([], dtype=int)
            keep = rng.random(nxt.size) < 0.9
            a[nxt[keep]] = 0x00                                 # ... most of them stuffing, at every alignment
            cases.append(a.tobytes())
    for b in cases:
        buf = np.frombuffer(b + b"\xff\xd9" * 8, dtype=np.uint8).copy()      # bytes behind the segment must not be looked at
        got = fn(buf.ctypes.data, len(b))
        assert got == naive(b), (len(b), got, naive(b))
