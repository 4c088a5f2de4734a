"""ctypes front-end of the CPU oracle (oracle/jpezy_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under jpezy_amd/ imports this module.  PARITY UNPINNED: see jpezy_oracle.h.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


def build(force=False, constants=None, out=None):
    """Compile libjpezy_oracle.so with gcc (plain IEEE-754 double, no contraction).  constants/out: build the same
    sources against an alternative constants header into another file (tests/test_constants_override.py)."""
    so = Path(out) if out else _HERE / "libjpezy_oracle.so"
    srcs = [_HERE / "jpezy_oracle.c", _HERE / "jpezy_oracle.h", _HERE.parent / "include" / "jpezy_constants.h"]
    if constants:
        srcs.append(Path(constants))
    if force or not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs if s.exists()):
        so.parent.mkdir(parents=True, exist_ok=True)
        cmd = ["make", "-C", str(_HERE), "-B", f"OUT={so}"]
        if constants:
            cmd.append(f"CONSTANTS={Path(constants).resolve()}")
        subprocess.check_call(cmd + [str(so)], stdout=subprocess.DEVNULL)
    return so


class FrameInfo(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("ncomp", C.c_int), ("precision", C.c_int),
        ("H", C.c_int * 3), ("V", C.c_int * 3), ("Tq", C.c_int * 3),
        ("hmax", C.c_int), ("vmax", C.c_int),
        ("mcu_cols", C.c_int), ("mcu_rows", C.c_int), ("blocks_per_mcu", C.c_int),
        ("restart_interval", C.c_int),
        ("major_rev", C.c_int), ("minor_rev", C.c_int), ("units", C.c_int),
        ("hdensity", C.c_int), ("vdensity", C.c_int), ("jfif", C.c_int),
        ("comment", C.c_char * 256),
        ("qt", (C.c_uint16 * 64) * 4),
    ]


def lib():
    global _LIB
    if _LIB is None:
        # JPEZY_ORACLE_LIB: another build of the oracle (alternative constants, tests/test_constants_override.py)
        so = Path(os.environ["JPEZY_ORACLE_LIB"]) if os.environ.get("JPEZY_ORACLE_LIB") else build()
        L = C.CDLL(str(so))
        u8p, i16p, ip = C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.POINTER(C.c_int)
        L.jo_rgb_y.argtypes = L.jo_rgb_cb.argtypes = L.jo_rgb_cr.argtypes = [C.c_uint8] * 3
        L.jo_fdct_block.argtypes = [ip, ip]
        L.jo_quantize_block.argtypes = [ip, C.c_int]
        L.jo_idct_block.argtypes = [ip, C.c_int, ip]
        L.jo_encode_coeffs_rows.argtypes = [u8p, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i16p]
        L.jo_encode_coeffs_rows.restype = None
        L.jo_write_jpeg.argtypes = [i16p, C.c_int, C.c_int, C.c_int, C.c_char_p, u8p, C.c_size_t]
        L.jo_write_jpeg.restype = C.c_long
        L.jo_read_jpeg.argtypes = [u8p, C.c_size_t, C.POINTER(FrameInfo), i16p, C.c_size_t]
        L.jo_decode_planes_rows.argtypes = [i16p, C.POINTER(FrameInfo), C.c_int, C.c_int, C.c_int, u8p, u8p, u8p]
        L.jo_decode_planes_rows.restype = None
        L.jo_read_ppm_p3.argtypes = [C.c_char_p, ip, ip, C.POINTER(u8p), C.POINTER(u8p), C.POINTER(u8p)]
        L.jo_format_ppm_p3.argtypes = [C.c_int, C.c_int, u8p, u8p, u8p, C.c_char_p, C.c_size_t]
        L.jo_format_ppm_p3.restype = C.c_long
        L.jo_free.argtypes = [C.c_void_p]
        L.jo_zz.restype = ip
        L.jo_qt.restype = ip
        L.jo_qt.argtypes = [C.c_int]
        L.jo_cos_table.restype = C.POINTER(C.c_double)
        L.jo_inv_sqrt2.restype = C.c_double
        _LIB = L
    return _LIB


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _i16(a):
    return a.ctypes.data_as(C.POINTER(C.c_int16))


def mcu_grid(W, H):
    return (W + 15) // 16, (H + 15) // 16


def _planes(r, g, b, W, H):
    out = []
    for p in (r, g, b):
        p = np.ascontiguousarray(p, dtype=np.uint8).reshape(-1)
        assert p.size == W * H
        out.append(p)
    return out


def encode_coeffs(r, g, b, W, H, gray=False, rows=None):
    """planar u8 r,g,b -> int16 [mcu_rows, mcu_cols, 6|4, 64] zig-zag coefficients."""
    r, g, b = _planes(r, g, b, W, H)
    mc, mr = mcu_grid(W, H)
    out = np.zeros((mr, mc, 4 if gray else 6, 64), dtype=np.int16)
    y0, y1 = (0, mr) if rows is None else rows
    lib().jo_encode_coeffs_rows(_u8(r), _u8(g), _u8(b), W, H, int(gray), y0, y1, _i16(out))
    return out


def write_jpeg(coeffs, W, H, gray=False, comment=None):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
    if comment is None:
        comment = b"Encoded by JPEZY" if gray else b"Encoded by jpezy"
    cap = max(W * H * 3, 10240) + coeffs.size * 4 + 4096
    buf = np.zeros(cap, dtype=np.uint8)
    n = lib().jo_write_jpeg(_i16(coeffs), W, H, int(gray), comment, _u8(buf), cap)
    if n < 0:
        raise RuntimeError("jo_write_jpeg failed")
    return buf[:n].tobytes()


def encode_jpeg(r, g, b, W, H, gray=False):
    return write_jpeg(encode_coeffs(r, g, b, W, H, gray), W, H, gray)


def read_jpeg(data):
    """.jpg bytes -> (FrameInfo, int16 coeffs [mcu_rows, mcu_cols, blocks_per_mcu, 64] zig-zag)."""
    arr = np.frombuffer(data, dtype=np.uint8).copy()
    info = FrameInfo()
    rc = lib().jo_read_jpeg(_u8(arr), arr.size, C.byref(info), None, 0)
    if rc:
        raise RuntimeError(f"jo_read_jpeg header rc={rc}")
    co = np.zeros((info.mcu_rows, info.mcu_cols, info.blocks_per_mcu, 64), dtype=np.int16)
    rc = lib().jo_read_jpeg(_u8(arr), arr.size, C.byref(info), _i16(co), co.size)
    if rc:
        raise RuntimeError(f"jo_read_jpeg scan rc={rc}")
    return info, co


def decode_planes(coeffs, info, gray=False, rows=None, out=None):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
    W, H = info.width, info.height
    if out is None:
        out = [np.zeros(W * H, dtype=np.uint8) for _ in range(3)]
    y0, y1 = (0, info.mcu_rows) if rows is None else rows
    lib().jo_decode_planes_rows(_i16(coeffs), C.byref(info), int(gray), y0, y1, _u8(out[0]), _u8(out[1]), _u8(out[2]))
    return out


def decode_jpeg(data, gray=False):
    info, co = read_jpeg(data)
    r, g, b = decode_planes(co, info, gray)
    return info, r, g, b


def make_info(W, H, gray_layout=False):
    """FrameInfo of a file jpezy_encode itself writes (2x2,1x1,1x1; Annex-K tables)."""
    info = FrameInfo()
    info.width, info.height, info.ncomp, info.precision = W, H, 3, 8
    for i, (h, v, tq) in enumerate([(2, 2, 0), (1, 1, 1), (1, 1, 1)]):
        info.H[i], info.V[i], info.Tq[i] = h, v, tq
    info.hmax = info.vmax = 2
    info.mcu_cols, info.mcu_rows = mcu_grid(W, H)
    info.blocks_per_mcu = 6
    L = lib()
    for t in range(2):
        q = L.jo_qt(t)
        for i in range(64):
            info.qt[t][i] = q[i]
    return info


def read_ppm_p3(path):
    L = lib()
    W, H = C.c_int(), C.c_int()
    pr, pg, pb = (C.POINTER(C.c_uint8)() for _ in range(3))
    rc = L.jo_read_ppm_p3(os.fsencode(path), C.byref(W), C.byref(H), C.byref(pr), C.byref(pg), C.byref(pb))
    if rc:
        raise RuntimeError(f"jo_read_ppm_p3 rc={rc}")
    n = W.value * H.value
    out = [np.ctypeslib.as_array(p, shape=(n,)).copy() for p in (pr, pg, pb)]
    for p in (pr, pg, pb):
        L.jo_free(p)
    return W.value, H.value, out[0], out[1], out[2]


def format_ppm_p3(W, H, r, g, b):
    r, g, b = _planes(r, g, b, W, H)
    cap = 64 + 12 * W * H
    buf = C.create_string_buffer(cap)
    n = lib().jo_format_ppm_p3(W, H, _u8(r), _u8(g), _u8(b), buf, cap)
    if n < 0:
        raise RuntimeError("jo_format_ppm_p3 failed")
    return buf.raw[:n]


def constants():
    L = lib()
    return {
        "zz": np.array([L.jo_zz()[i] for i in range(64)]),
        "qt_luma": np.array([L.jo_qt(0)[i] for i in range(64)]),
        "qt_chroma": np.array([L.jo_qt(1)[i] for i in range(64)]),
        "cos": np.array([L.jo_cos_table()[i] for i in range(64)]),
        "inv_sqrt2": L.jo_inv_sqrt2(),
    }


def synth_rgb(W, H, seed=0x6A70657A79, frame=0):
    """iid uniform u8 planes from a counter-based SplitMix64 stream: value k of the stream
    (seed, frame) is byte 0 of splitmix64(seed + frame*2^40 + k); pixel p, channel c uses k = 3p + c."""
    n = W * H * 3
    k = np.arange(n, dtype=np.uint64) + np.uint64((seed + (frame << 40)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        z = k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    v = (z & np.uint64(0xFF)).astype(np.uint8).reshape(H * W, 3)
    return np.ascontiguousarray(v[:, 0]), np.ascontiguousarray(v[:, 1]), np.ascontiguousarray(v[:, 2])
