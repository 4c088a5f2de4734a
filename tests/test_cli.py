"""jpezy_encode / jpezy_decode (C++ host side over the C-ABI): argv rules, exit codes and transcript of the
reference's CLIs (src/encoder/main.cpp, src/decoder/main.cpp) on CPU; byte-exact files on the GPU."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "jpezy_amd" / "bin"


@pytest.fixture(scope="module")
def cli():
    from jpezy_amd import _build
    _build.build_all()
    enc, dec = BIN / "jpezy_encode", BIN / "jpezy_decode"
    assert enc.exists() and dec.exists()
    return enc, dec


def _run(*args):
    return subprocess.run([str(a) for a in args], capture_output=True, text=True, timeout=300)


def _write_ppm(oracle, path, W, H, frame=0):
    r, g, b = oracle.synth_rgb(W, H, frame=frame)
    # the reference's own writers put one "r g b" triple per line; encode_io accepts any token layout
    path.write_bytes(oracle.format_ppm_p3(W, H, r, g, b))
    return r, g, b


def test_usage_and_exit_codes(cli, tmp_path):
    enc, dec = cli
    p = _run(enc)
    assert p.returncode == 1 and p.stderr.startswith("Usage: jpezy_encode <input.ppm>")
    p = _run(enc, tmp_path / "a.ppm", "out.png")                 # unknown output kind: usage, before the logo
    assert p.returncode == 1 and "Usage: jpezy_encode" in p.stderr and "by roki" not in p.stdout
    p = _run(enc, tmp_path / "missing.ppm", tmp_path / "o.jpg")  # unreadable input
    assert p.returncode == 1 and "The file is not found or the formatting error" in p.stderr and "by roki" in p.stdout
    p = _run(dec, "x.png", "y.ppm")
    assert p.returncode == 1 and p.stderr.startswith("Usage: jpezy_decode <input.(jpg | jpeg)>")
    p = _run(dec, tmp_path / "missing.jpg", tmp_path / "y.ppm")
    assert p.returncode == 1 and "decode failed" in p.stderr and "process started..." in p.stdout


def test_ppm_passthrough_and_parser_quirks(cli, oracle, tmp_path):
    """Mode::PPM / --debug need no GPU: they exercise the P3 reader and writer (encode_io.hpp:45-119)."""
    enc, _ = cli
    W, H = 9, 4
    src = tmp_path / "in.ppm"
    r, g, b = _write_ppm(oracle, src, W, H)
    out = tmp_path / "copy.ppm"
    p = _run(enc, src, out)
    assert p.returncode == 0, p.stderr
    assert "Reading the input file... width: 9 height: 4" in p.stdout and "Total processing time:" in p.stdout
    lines = out.read_text().split("\n")
    assert lines[:3] == ["P3", "9 4", "255"]
    assert lines[3] == f"{r[0]} {g[0]} {b[0]}" and len(lines) == 3 + W * H + 1
    p = _run(enc, src, "--debug")
    assert p.returncode == 0 and "P3\n9 4\n255\n" in p.stdout
    # any line containing '#' is skipped; several pixels per line are fine
    body = " ".join(f"{r[i]} {g[i]} {b[i]}" for i in range(W * H))
    src.write_text(f"P3\n# made by a test\n{W} {H}\n255\n{body}\n")
    assert _run(enc, src, out).returncode == 0 and out.read_text().split("\n")[3] == f"{r[0]} {g[0]} {b[0]}"
    # width/height must be one line of exactly two tokens; "P3" must be alone on its line
    src.write_text(f"P3\n{W}\n{H}\n255\n{body}\n")
    assert _run(enc, src, out).returncode == 1
    src.write_text(f"P3 \n{W} {H}\n255\n{body}\n")
    assert _run(enc, src, out).returncode == 1
    # a last line without trailing newline is dropped -> too few pixels -> error instead of the reference's OOB read
    src.write_text(f"P3\n{W} {H}\n255\n{body}")
    assert _run(enc, src, out).returncode == 1


def test_encode_without_gpu_fails_loudly(cli, oracle, tmp_path):
    import jpezy_amd as J
    if J.load_library().jpezy_hip_device_count() > 0:
        pytest.skip("a HIP device is present")
    enc, _ = cli
    src = tmp_path / "in.ppm"
    _write_ppm(oracle, src, 16, 16)
    p = _run(enc, src, tmp_path / "o.jpg")
    assert p.returncode == 1 and "no HIP device" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(512, 512), (100, 37)])
def test_cli_roundtrip_is_byte_exact(cli, oracle, tmp_path, size):
    """BASELINE configs[0] (plumbing): PPM -> jpezy_encode -> .jpg -> jpezy_decode -> PPM, every file compared
    with what the oracle (the reference's algorithm on the CPU) produces from the same input."""
    enc, dec = cli
    W, H = size
    src = tmp_path / "in.ppm"
    r, g, b = _write_ppm(oracle, src, W, H, frame=21)
    for gray in (False, True):
        jpg = tmp_path / ("g.jpg" if gray else "c.jpg")
        p = _run(enc, src, jpg, *(["--gray"] if gray else []))
        assert p.returncode == 0, p.stderr
        want = oracle.encode_jpeg(r, g, b, W, H, gray=gray)
        data = jpg.read_bytes()
        assert data == want
        unit = "srook::byte" if gray else "byte"                    # sic, encode_io.hpp:166,193
        assert f"Output size: {len(want)} {unit}" in p.stdout
        for key in ["Write JPEG Header ...", "Encoding ...", "Write EOI ...", "Start encoding and writing ..."]:
            assert key in p.stdout
        for dgray in (False, True):
            ppm = tmp_path / "out.ppm"
            p = _run(dec, jpg, ppm, *(["--gray"] if dgray else []))
            assert p.returncode == 0, p.stderr
            info, dr, dg, db = oracle.decode_jpeg(data, gray=dgray)
            assert ppm.read_bytes() == oracle.format_ppm_p3(W, H, dr, dg, db)
            comment = "Encoded by JPEZY" if gray else "Encoded by jpezy"
            assert f'Loaded JPEG: {W}x{H}, presicion 8, "{comment}", JFIF standart 1.02, dots inch, frames 3, density 96x96' in p.stdout
            assert f"Decoded image: Netpbm image data, size = {W} x {H}, pixmap, ASCII text" in p.stdout
    p = _run(dec, tmp_path / "c.jpg", tmp_path / "v.ppm", "-v")
    assert p.returncode == 0
    for m in ["[APP0]", "[COM]", "[DQT]", "[DHT]", "[SOF0]", "[SOS]"]:
        assert f"found marker: {m}" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["444", "422", "gray1"])
def test_cli_decodes_foreign_layouts(cli, oracle, tmp_path, kind):
    """files jpezy_encode never writes (4:4:4, 4:2:2, one component) decode through the generic kernels, like the
    reference's general decode_mcu loop does on the CPU -- compared with the oracle byte for byte"""
    from PIL import Image
    _, dec = cli
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (45, 70, 3), dtype=np.uint8)
    p = tmp_path / "x.jpg"
    if kind == "gray1":
        Image.fromarray(img[..., 0]).save(p, "JPEG", quality=80)
    else:
        Image.fromarray(img).save(p, "JPEG", quality=80, subsampling=0 if kind == "444" else 1)
    for dgray in (False, True):
        out = tmp_path / "x.ppm"
        r = _run(dec, p, out, *(["--gray"] if dgray else []))
        assert r.returncode == 0, r.stderr
        info, dr, dg, db = oracle.decode_jpeg(p.read_bytes(), gray=dgray)
        assert out.read_bytes() == oracle.format_ppm_p3(70, 45, dr, dg, db)
