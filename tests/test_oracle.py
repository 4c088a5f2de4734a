"""CPU tests of the oracle (oracle/jpezy_oracle.c): constants, the hazards SURVEY.md names, the committed
fixtures, and independent cross-checks (libjpeg via PIL, /usr/bin/file).  PARITY UNPINNED: the reference
has no vectors, so "golden" here means frozen oracle outputs (tests/golden/gen_golden.py)."""
import ctypes as C
import hashlib
import io
import json
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
FIXTURES = sorted(p.stem for p in (ROOT / "tests" / "golden").glob("*.npz"))


def test_zigzag_is_the_annex_a_scan(oracle):
    zz = oracle.constants()["zz"]
    assert sorted(zz.tolist()) == list(range(64))
    # consecutive entries are neighbours on an anti-diagonal walk; first entries as in jpezy.hpp:36-38
    assert zz[:12].tolist() == [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25]
    rc = [(int(v) // 8, int(v) % 8) for v in zz]
    sums = [r + c for r, c in rc]
    assert sums == sorted(sums)


def test_quant_tables_are_annex_k(oracle):
    c = oracle.constants()
    assert c["qt_luma"][:8].tolist() == [16, 11, 10, 16, 24, 40, 51, 61]
    assert c["qt_luma"][-8:].tolist() == [72, 92, 95, 98, 112, 100, 103, 99]
    assert c["qt_chroma"][:8].tolist() == [17, 18, 24, 47, 99, 99, 99, 99]
    assert int(c["qt_luma"].sum()) == 3688 and int(c["qt_chroma"].sum()) == 5505  # sums of jpezy.hpp:131-152


def test_cos_table_is_correctly_rounded(oracle):
    sys.path.insert(0, str(ROOT / "tools" / "gen"))
    import gen_constants
    want = gen_constants.cos_table()
    got = oracle.constants()["cos"]
    assert [float(x).hex() for x in got] == [float(x).hex() for x in want]
    # index convention [u*8+x] = cos((2x+1)u*pi/16) (ref jpezy_encoder.hpp:160), exact symmetries
    t = got.reshape(8, 8)
    assert np.all(t[0] == 1.0)
    for u in range(8):
        assert np.array_equal(t[u, ::-1], (-1) ** u * t[u])
    assert np.allclose(t, [[np.cos((2 * x + 1) * u * np.pi / 16) for x in range(8)] for u in range(8)], atol=1e-15, rtol=0)
    assert float(oracle.constants()["inv_sqrt2"]).hex() == "0x1.6a09e667f3bccp-1"


def test_truncating_colour_conversion_hazard(oracle):
    """for grey pixels Y != c-128 for exactly 30 of 256 levels (truncation of a double sum)."""
    L = oracle.lib()
    bad = [c for c in range(256) if L.jo_rgb_y(c, c, c) != c - 128]
    assert len(bad) == 30 and bad[:6] == [143, 149, 156, 157, 162, 169]
    # chroma is zero-centred and truncates toward zero
    assert L.jo_rgb_cb(255, 0, 0) == -43 and L.jo_rgb_cr(255, 0, 0) == 127 and L.jo_rgb_cb(0, 0, 255) == 127


def test_flat_block_dc_hazard(oracle):
    """S*S = 0.4999999999999999 makes a flat block of level c give DC = 8c -/+ 1 toward zero."""
    L = oracle.lib()
    off = 0
    for c in range(-128, 128):
        pic = (C.c_int * 64)(*([c] * 64))
        out = (C.c_int * 64)()
        L.jo_fdct_block(pic, out)
        want = 8 * c - (1 if c > 0 else -1 if c < 0 else 0)
        assert out[0] == want
        assert all(out[k] == 0 for k in range(1, 64))
        off += out[0] != 8 * c
    assert off == 255


def test_fdct_matches_a_float128_style_evaluation(oracle):
    """the oracle's DCT is the textbook 2-D DCT-II: compare against numpy longdouble away from boundaries"""
    L = oracle.lib()
    rng = np.random.default_rng(1)
    ld = np.longdouble
    cosv = np.array([[np.cos(ld((2 * x + 1) * u) * np.pi / 16) for x in range(8)] for u in range(8)], dtype=ld)
    for _ in range(20):
        pic = rng.integers(-128, 128, 64)
        out = (C.c_int * 64)()
        L.jo_fdct_block((C.c_int * 64)(*pic.tolist()), out)
        P = pic.reshape(8, 8).astype(ld)
        F = cosv @ P @ cosv.T
        for i in range(8):
            for j in range(8):
                v = F[i, j] * (1 / np.sqrt(ld(2)) if i == 0 else 1) * (1 / np.sqrt(ld(2)) if j == 0 else 1) / 4
                if abs(v - np.round(v)) > 1e-6:
                    assert out[i * 8 + j] == int(np.trunc(v))


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_reproduces_golden(oracle, golden_dir, name):
    z = np.load(golden_dir / f"{name}.npz")
    W, H = int(z["W"]), int(z["H"])
    co = oracle.encode_coeffs(z["r"], z["g"], z["b"], W, H, False)
    cog = oracle.encode_coeffs(z["r"], z["g"], z["b"], W, H, True)
    assert np.array_equal(co, z["coeffs"]) and np.array_equal(cog, z["coeffs_gray"])
    assert np.array_equal(co[:, :, :4], cog)           # GRAY_MODE only drops chroma (jpezy_encoder.hpp:61-64)
    assert oracle.write_jpeg(co, W, H, False) == z["jpg"].tobytes()
    assert oracle.write_jpeg(cog, W, H, True) == z["jpg_gray"].tobytes()
    info, rco = oracle.read_jpeg(z["jpg"].tobytes())
    assert (info.width, info.height, info.ncomp, info.precision) == (W, H, 3, 8)
    assert np.array_equal(rco, co)
    r, g, b = oracle.decode_planes(rco, info, False)
    assert np.array_equal(r, z["dec_r"]) and np.array_equal(g, z["dec_g"]) and np.array_equal(b, z["dec_b"])
    gr, gg, gb = oracle.decode_planes(rco, info, True)
    assert np.array_equal(gr, z["dec_gray"]) and np.array_equal(gg, gr) and np.array_equal(gb, gr)
    # a gray file still carries (all-zero) chroma blocks, 4 bits each
    _, gco = oracle.read_jpeg(z["jpg_gray"].tobytes())
    assert np.array_equal(gco[:, :, :4], cog) and not gco[:, :, 4:].any()


def test_digests(oracle, golden_dir):
    d = json.loads((golden_dir / "digests.json").read_text())["rand512"]
    r, g, b = oracle.synth_rgb(d["W"], d["H"], frame=d["frame"])
    co = oracle.encode_coeffs(r, g, b, d["W"], d["H"])
    assert hashlib.sha256(co.tobytes()).hexdigest() == d["coeffs_sha256"]
    jpg = oracle.write_jpeg(co, d["W"], d["H"])
    assert len(jpg) == d["jpg_len"] and hashlib.sha256(jpg).hexdigest() == d["jpg_sha256"]


def test_header_bytes(oracle):
    """exact fixed header; 644 bytes before the entropy data for the 16-char comment."""
    W, H = 48, 32
    r, g, b = oracle.synth_rgb(W, H)
    jpg = oracle.encode_jpeg(r, g, b, W, H)
    assert jpg[:20] == bytes.fromhex("ffd8ffe000104a46494600010201006000600000")
    assert jpg[20:24] == bytes.fromhex("fffe0013") and jpg[24:41] == b"Encoded by jpezy\0"
    assert jpg[41:46] == bytes.fromhex("ffdb004300") and jpg[46] == 16 and jpg[47] == 11 and jpg[48] == 12
    sos = jpg.index(bytes.fromhex("ffda000c03"))
    assert jpg[sos - 19:sos] == bytes.fromhex("ffc0001108") + bytes([0, H, 0, W]) + bytes.fromhex("03002200011101021101")
    assert jpg[sos:sos + 14] == bytes.fromhex("ffda000c03000001110211003f00")
    assert sos + 14 == 644
    assert jpg[-2:] == b"\xff\xd9"
    gj = oracle.encode_jpeg(r, g, b, W, H, gray=True)
    assert gj[24:41] == b"Encoded by JPEZY\0"                 # encode_io.hpp:181


def test_gray_chroma_blocks_cost_four_bits(oracle):
    """flat-zero chroma = DC cat 0 '00' + EOB '00'; an all-zero luma block = '00' + '1010'."""
    W, H = 16, 16
    z = np.zeros(W * H, np.uint8) + 128          # Y = 0 exactly for level 128
    assert oracle.lib().jo_rgb_y(128, 128, 128) == 0
    jpg = oracle.encode_jpeg(z, z, z, W, H, gray=True)
    ent = jpg[644:-2]
    # 4 luma blocks x 6 bits + 2 chroma x 4 bits = 32 bits = 4 bytes: 00 1010 | 00 1010 | 00 1010 | 00 1010 | 0000 | 0000
    assert ent == bytes([0b00101000, 0b10100010, 0b10001010, 0b00000000])


@pytest.mark.parametrize("size", [(64, 48), (33, 17), (160, 96), (16, 16)])
def test_libjpeg_decodes_oracle_files(oracle, size):
    """Independent pin of the bit stream: libjpeg (PIL) must parse the file, and its luma (rounded IDCT)
    equals the oracle's luma (truncating IDCT) or is one more."""
    from PIL import Image
    W, H = size
    r, g, b = oracle.synth_rgb(W, H, frame=11)
    jpg = oracle.encode_jpeg(r, g, b, W, H)
    im = Image.open(io.BytesIO(jpg))
    im.draft("YCbCr", (W, H))
    im.load()
    assert im.size == (W, H) and im.mode == "YCbCr"
    luma = np.asarray(im)[..., 0].astype(int)
    _, yr, _, _ = oracle.decode_jpeg(jpg, gray=True)
    d = luma - yr.reshape(H, W).astype(int)
    assert d.min() >= 0 and d.max() <= 1


def test_oracle_reads_libjpeg_files(oracle):
    """decode side: a 4:2:0 baseline file written by libjpeg decodes to within truncation of libjpeg's own
    luma; a 4:4:4 and a 1-component file exercise the general sampling loop (decode_mcu :504-528)."""
    from PIL import Image
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    for kw, ncomp in [(dict(subsampling=2), 3), (dict(subsampling=0), 3), (dict(), 1)]:
        buf = io.BytesIO()
        src = Image.fromarray(img) if ncomp == 3 else Image.fromarray(img[..., 0])
        src.save(buf, "JPEG", quality=85, **kw)
        info, r, g, b = oracle.decode_jpeg(buf.getvalue(), gray=True)
        assert (info.width, info.height, info.ncomp) == (56, 40, ncomp)
        im = Image.open(io.BytesIO(buf.getvalue()))
        if ncomp == 3:
            im.draft("YCbCr", (56, 40))
        im.load()
        a = np.asarray(im)
        luma = (a[..., 0] if ncomp == 3 else a).astype(int)
        d = luma - r.reshape(40, 56).astype(int)
        assert d.min() >= 0 and d.max() <= 1


@pytest.mark.skipif(shutil.which("file") is None, reason="/usr/bin/file not present")
def test_file_utility_describes_the_header(oracle, tmp_path):
    """README.md:59 is the reference's only recorded expected output for the header."""
    W, H = 512, 512
    r, g, b = oracle.synth_rgb(W, H)
    p = tmp_path / "o.jpg"
    p.write_bytes(oracle.encode_jpeg(r, g, b, W, H))
    out = subprocess.run(["file", str(p)], capture_output=True, text=True).stdout
    for field in ["JPEG image data", "JFIF standard 1.02", "resolution (DPI)", "density 96x96", "segment length 16",
                  'comment: "Encoded by jpezy"', "baseline", "precision 8", "512x512"]:
        assert field in out, (field, out)
    assert "frames 3" in out or "components 3" in out


def test_ppm_roundtrip(oracle, tmp_path):
    W, H = 5, 3
    r, g, b = oracle.synth_rgb(W, H)
    text = oracle.format_ppm_p3(W, H, r, g, b)
    assert text.startswith(b"P3\n# Decoded by jpezy\n5 3\n255\n") and text.count(b"\n") == 4 + W * H
    p = tmp_path / "a.ppm"
    p.write_bytes(text)
    W2, H2, r2, g2, b2 = oracle.read_ppm_p3(p)
    assert (W2, H2) == (W, H) and np.array_equal(r, r2) and np.array_equal(g, g2) and np.array_equal(b, b2)
    # quirk (encode_io.hpp:80): a last line without a trailing newline is dropped -> too few pixels
    p.write_bytes(text[:-1])
    with pytest.raises(RuntimeError):
        oracle.read_ppm_p3(p)
