"""CPU tests of the product's host side through the C-ABI (no compute calls: there is no GPU here):
the library loads and exports every symbol include/jpezy_hip.h declares, the Huffman/JFIF tail and the
marker-parser/Huffman head agree byte for byte with the oracle, and errors surface as statuses."""
import ctypes as C
import io
import re
from pathlib import Path

import numpy as np
import pytest

import jpezy_amd as J
from jpezy_amd import api

ROOT = Path(__file__).resolve().parent.parent
FIXTURES = sorted(p.stem for p in (ROOT / "tests" / "golden").glob("*.npz"))


def test_library_exports_every_declared_symbol():
    hdr = (ROOT / "include" / "jpezy_hip.h").read_text()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(jpezy_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = C.CDLL(str(J.library_path()))
    for name in sorted(declared):
        assert hasattr(lib, name), f"libjpezy_hip.so does not export {name}"
    assert declared == {n for n, _, _ in api.ABI}, "jpezy_amd.api.ABI is out of sync with include/jpezy_hip.h"


def test_no_cpu_fallback():
    lib = J.load_library()
    if lib.jpezy_hip_device_count() == 0:
        with pytest.raises(J.JpezyError, match="no HIP device"):
            J.Context(0)
    assert lib.jpezy_ctx_sync(None) < 0 and b"null context" in lib.jpezy_hip_last_error()


def test_geometry_helpers():
    assert J.mcu_grid(4096, 4096) == (256, 256)
    assert J.mcu_grid(1920, 1080) == (120, 68)       # height padded to 1088 by edge replication
    assert J.mcu_grid(17, 33) == (2, 3)
    assert J.coeff_count(4096, 4096) == 25165824 and J.coeff_count(7680, 4320, True) == 33177600
    assert J.coeff_count(0, 5) == 0


@pytest.mark.parametrize("name", FIXTURES)
def test_write_and_read_match_golden(golden_dir, name):
    z = np.load(golden_dir / f"{name}.npz")
    W, H = int(z["W"]), int(z["H"])
    assert J.write_jpeg(z["coeffs"], W, H, False) == z["jpg"].tobytes()
    assert J.write_jpeg(z["coeffs_gray"], W, H, True) == z["jpg_gray"].tobytes()
    info, co = J.read_jpeg(z["jpg"].tobytes())
    assert (info.width, info.height, info.ncomp, info.precision, info.hmax, info.vmax) == (W, H, 3, 8, 2, 2)
    assert [info.H[i] for i in range(3)] == [2, 1, 1] and [info.Tq[i] for i in range(3)] == [0, 1, 1]
    assert (info.format, info.major_rev, info.minor_rev, info.units, info.hdensity, info.vdensity) == (1, 1, 2, 1, 96, 96)
    assert info.comment == b"Encoded by jpezy"
    assert np.array_equal(co, z["coeffs"])


def _stress_coeffs(rng, nmcu):
    """blocks that reach every branch of encode_huffman: long zero runs (ZRL), values up to the 10-bit
    category, a non-zero last coefficient (no EOB), all-zero blocks, big DC swings."""
    co = np.zeros((nmcu, 6, 64), np.int16)
    for m in range(nmcu):
        for b in range(6):
            kind = rng.integers(0, 6)
            blk = co[m, b]
            if kind == 0:
                pass
            elif kind == 1:
                blk[:] = rng.integers(-1023, 1024, 64)
            elif kind == 2:
                blk[0] = rng.integers(-1000, 1000)
                blk[63] = rng.integers(1, 1024)
            elif kind == 3:
                idx = rng.choice(63, size=3, replace=False) + 1
                blk[idx] = rng.integers(-300, 300, 3)
            elif kind == 4:
                blk[0] = rng.integers(-1023, 1024)
                blk[40] = -1
            else:
                blk[:] = rng.integers(-3, 4, 64)
    return co


def test_write_jpeg_matches_oracle_on_stress_coefficients(oracle):
    rng = np.random.default_rng(42)
    for W, H in [(16, 16), (48, 32), (100, 60)]:
        mc, mr = J.mcu_grid(W, H)
        co = _stress_coeffs(rng, mc * mr).reshape(mr, mc, 6, 64)
        a = oracle.write_jpeg(co, W, H, False)
        assert J.write_jpeg(co, W, H, False) == a
        info, back = J.read_jpeg(a)
        oinfo, oback = oracle.read_jpeg(a)
        assert np.array_equal(back, co) and np.array_equal(oback, co)
        g = co[:, :, :4].copy()
        assert J.write_jpeg(g, W, H, True) == oracle.write_jpeg(g, W, H, True)
        assert J.write_jpeg(co, W, H, False, comment=b"") == oracle.write_jpeg(co, W, H, False, comment=b"")


def test_byte_stuffing_and_zero_padding():
    """0xFF entropy bytes get a stuffed 0x00; the last partial byte is padded with ZERO bits."""
    co = np.zeros((1, 1, 6, 64), np.int16)
    co[0, 0, 0, 0] = -1023      # luma DC cat 10: code 11111110 then 0000000000 -> starts with 0xFE..
    co[0, 0, 0, 1:] = 1023      # many 1-bits: guarantees 0xFF bytes in the stream
    jpg = J.write_jpeg(co, 16, 16)
    ent = jpg[644:-2]
    assert b"\xff" in ent
    for i, byte in enumerate(ent[:-1]):
        if byte == 0xFF:
            assert ent[i + 1] == 0x00
    info, back = J.read_jpeg(jpg)
    assert np.array_equal(back, co)
    # all-zero gray MCU: 32 entropy bits exactly (see tests/test_oracle.py); 17x17 -> 4 MCUs, still byte exact
    z = np.zeros((2, 2, 4, 64), np.int16)
    assert J.write_jpeg(z, 17, 17, True)[644:-2] == bytes([0x28, 0xA2, 0x8A, 0x00] * 4)
    z[1, 1, 3, 0] = 1           # one DC diff of 1 in the last luma block: '010' + '1' then chroma blocks
    ent = J.write_jpeg(z, 17, 17, True)[644:-2]
    assert len(ent) == 17 and ent[-1] & 0x0F == 0   # trailing pad bits are zeros


def test_out_of_table_coefficients_are_an_error():
    co = np.zeros((1, 1, 6, 64), np.int16)
    co[0, 0, 0, 5] = 1024       # needs an 11-bit AC category: not in K.5 (reference: out-of-range index)
    with pytest.raises(J.JpezyError):
        J.write_jpeg(co, 16, 16)
    co[0, 0, 0, 5] = 0
    co[0, 0, 0, 0] = 2048       # DC category 12
    with pytest.raises(J.JpezyError):
        J.write_jpeg(co, 16, 16)


def test_read_jpeg_matches_oracle_on_libjpeg_files(oracle):
    from PIL import Image
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (72, 104, 3), dtype=np.uint8)
    variants = [dict(subsampling=2, quality=50), dict(subsampling=0, quality=90), dict(subsampling=1, quality=75),
                dict(subsampling=2, quality=30, optimize=True), dict(subsampling=2, quality=60, restart_marker_blocks=3)]
    for kw in variants:
        buf = io.BytesIO()
        try:
            Image.fromarray(img).save(buf, "JPEG", **kw)
        except TypeError:
            continue
        data = buf.getvalue()
        oi, oc = oracle.read_jpeg(data)
        info, co = J.read_jpeg(data)
        assert (info.width, info.height, info.ncomp, info.blocks_per_mcu, info.restart_interval) == \
               (oi.width, oi.height, oi.ncomp, oi.blocks_per_mcu, oi.restart_interval), kw
        assert bytes(info.qt) == bytes(oi.qt)
        assert np.array_equal(co, oc), kw
    buf = io.BytesIO()
    Image.fromarray(img[..., 1]).save(buf, "JPEG", quality=70)
    oi, oc = oracle.read_jpeg(buf.getvalue())
    info, co = J.read_jpeg(buf.getvalue())
    assert info.ncomp == 1 and np.array_equal(co, oc)


ODD_LAYOUTS = {
    "411": [(4, 1, 0, 0), (1, 1, 1, 1), (1, 1, 1, 1)],
    "h4v2_partial": [(4, 2, 0, 0), (2, 1, 1, 1), (2, 2, 1, 1)],           # H = 2 under hmax = 4: decode_mcu's overlapping writes
    "h3_partial": [(3, 1, 0, 0), (1, 1, 1, 1), (2, 1, 1, 0)],             # hmax % H != 0: the end of the plane is never written
    "v4": [(1, 4, 0, 0), (1, 2, 1, 1), (1, 3, 0, 1)],
    "h4v4": [(4, 4, 0, 0), (3, 3, 1, 1), (2, 2, 1, 1)],                   # 29 blocks per MCU
    "one_comp_2x2": [(2, 2, 0, 0)],
}


@pytest.mark.parametrize("name", sorted(ODD_LAYOUTS))
def test_read_jpeg_of_layouts_libjpeg_does_not_write(oracle, name):
    """sampling factors up to 4 and factors that do not divide hmax/vmax (ref decoder/jpezy_decoder.hpp:279-305 takes any
    nibble): the host head delivers the synthesised coefficients, and so does the oracle's restatement"""
    from jpeg_synth import synth_jpeg
    data, co, _ = synth_jpeg(101, 70, ODD_LAYOUTS[name], seed=len(name))
    oinfo, oco = oracle.read_jpeg(data)
    info, hco = api.read_jpeg(data)
    assert np.array_equal(np.asarray(oco).reshape(-1), co)
    assert np.array_equal(np.asarray(hco).reshape(-1), co)
    assert (info.hmax, info.vmax, info.blocks_per_mcu) == (oinfo.hmax, oinfo.vmax, oinfo.blocks_per_mcu)


def test_malformed_streams_fail_cleanly(golden_dir):
    z = np.load(golden_dir / "rand64.npz")
    jpg = z["jpg"].tobytes()
    for bad in [b"", b"\x00" * 64, jpg[:100], jpg[:700], jpg[:2] + b"\xff\xd9", jpg.replace(b"\xff\xc0", b"\xff\xc2", 1)]:
        with pytest.raises(J.JpezyError):
            J.read_jpeg(bad)


def test_write_jpeg_batch_is_threaded_and_byte_identical(oracle):
    import time
    W, H, F = 160, 112, 12
    co = np.stack([oracle.encode_coeffs(*oracle.synth_rgb(W, H, frame=f), W, H) for f in range(F)])
    want = [oracle.write_jpeg(co[f], W, H) for f in range(F)]
    for threads in (1, 4, 0):
        got = J.write_jpeg_batch(co, W, H, F, threads=threads)
        assert got == want
    bad = co.copy()
    bad[3, 0, 0, 0, 5] = 2000          # outside K.5: that frame fails, the call reports it
    with pytest.raises(J.JpezyError):
        J.write_jpeg_batch(bad, W, H, F)


def test_host_codec_under_asan_ubsan_on_mutated_files():
    """tests/fuzz/run_host_fuzz.py: the marker parser, Huffman reader and writer built with AddressSanitizer + UBSan (g++, CPU)
    and fed mutated files -- bit flips, truncations, insertions, header field edits.  Any out-of-bounds access fails the run.
    (Found in round 1: an over-subscribed DHT overran the 8-bit lookup table; a scan that selects a table no DHT defined
    read an uninitialised one.)"""
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tests" / "fuzz" / "run_host_fuzz.py"), "1500"], capture_output=True, text=True, timeout=900)
    if "cannot find -lasan" in r.stderr or "libasan" in r.stderr and "No such file" in r.stderr:
        pytest.skip("no libasan for g++ in this image")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "decoded" in r.stdout
    # the core of the GPU Huffman decoder (two-level tables, table builder, decode step: jpezy_amd/csrc/jpezy_huffdec_core.h) walked on
    # the CPU over the same files under the same sanitizers (tests/fuzz/huffdec_core_fuzz.cpp): what it decodes is what the host decoder
    # decodes, and no encoder's file is declined
    m = re.search(r"decode step on the CPU: (\d+) files walked and equal to the host decoder", r.stdout)
    assert m and int(m.group(1)) >= 100, r.stdout
    # ... and every table of those files, and 3,000 random canonical codes, decode each of the 65,536 16-bit windows as the canonical
    # code says, through the two independent lookups of round 4; what the builder declines it cannot express
    m = re.search(r"(\d+) tables of files and (\d+) random ones \((\d+) declined", r.stdout)
    assert m and int(m.group(1)) >= 400 and int(m.group(2)) == 3000 and 0 < int(m.group(3)) < 3000, r.stdout


def test_scan_that_selects_an_undefined_table_is_an_error(golden_dir):
    """the reference's huffman_table of a slot no DHT filled is empty: decode_huffman_impl finds no code and throws
    (decoder/jpezy_decoder.hpp:629-641)"""
    z = np.load(sorted(golden_dir.glob("*.npz"))[0])
    key = [k for k in z.files if k.startswith("jpg")][0]
    d = bytearray(z[key].tobytes())
    sos = d.index(b"\xFF\xDA")
    assert d[sos + 4] == 3
    d[sos + 6] = 0x22            # first scan component: Td = Ta = 2, never defined
    with pytest.raises(api.JpezyError):
        api.read_jpeg(bytes(d))
    for _ in range(3):           # and stays an error whatever was decoded before (the table used to be uninitialised memory)
        api.read_jpeg(z[key].tobytes())
        with pytest.raises(api.JpezyError):
            api.read_jpeg(bytes(d))
