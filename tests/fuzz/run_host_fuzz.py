#!/usr/bin/env python3
"""Build tests/fuzz/host_codec_fuzz.cpp with AddressSanitizer + UBSan (g++, CPU) and run it over mutated .jpg files:

    python tests/fuzz/run_host_fuzz.py [n_mutants]

Seeds: tests/golden/*.jpg-like fixtures inside the .npz files, files written by PIL/libjpeg in several layouts, the
synthesised odd layouts of tests/jpeg_synth.py.  Mutations: bit flips, byte overwrites, truncations, insertions and
deletions, 16-bit field edits in the header.  Then the CLI's P3 reader on mutated text files (tests/fuzz/ppm_fuzz.cpp).  Any sanitizer report fails the run (exit code != 0).
"""
import io
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests"))


def seeds():
    from PIL import Image
    from jpeg_synth import synth_jpeg
    out = []
    for f in sorted((ROOT / "tests" / "golden").glob("*.npz")):
        z = np.load(f)
        for k in z.files:
            if k.startswith("jpg"):
                out.append(bytes(z[k].tobytes()))
    rng = np.random.default_rng(3)
    for kw in (dict(subsampling=0, quality=90), dict(subsampling=1, quality=60), dict(subsampling=2, quality=35, optimize=True),
               dict(subsampling=2, quality=70, restart_marker_blocks=2), dict(subsampling=0, quality=98, optimize=True),
               dict(subsampling=1, quality=96)):
        img = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", **kw)
        out.append(buf.getvalue())
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, (33, 47), dtype=np.uint8)).save(buf, "JPEG", quality=80)
    out.append(buf.getvalue())
    for comps in ([(4, 1, 0, 0), (1, 1, 1, 1), (1, 1, 1, 1)], [(3, 1, 0, 0), (1, 1, 1, 1), (2, 1, 1, 0)], [(2, 2, 0, 0)]):
        out.append(synth_jpeg(70, 40, comps, seed=1)[0])
    return out


def mutate(d, rng):
    b = bytearray(d)
    for _ in range(int(rng.integers(1, 6))):
        op = int(rng.integers(0, 6))
        n = len(b)
        if n < 8:
            break
        hdr = rng.random() < 0.5                       # half of the edits land in the first 700 bytes (markers, tables)
        pos = int(rng.integers(0, min(n, 700))) if hdr else int(rng.integers(0, n))
        if op == 0:
            b[pos] ^= 1 << int(rng.integers(0, 8))
        elif op == 1:
            b[pos] = int(rng.integers(0, 256))
        elif op == 2:
            del b[int(rng.integers(n // 2, n)):]
        elif op == 3:
            b[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
        elif op == 4:
            del b[pos:pos + int(rng.integers(1, 9))]
        else:
            v = int(rng.choice([0, 1, 2, 0xFF, 0xFFFF, 0x7FFF, 0x8000, 17]))
            b[pos:pos + 2] = v.to_bytes(2, "big")
    return bytes(b)


def ppm_phase(n):
    """the CLI's P3 reader (jpezy::encode_io) under the same sanitizers, on mutated text files"""
    exe = Path(tempfile.gettempdir()) / "jpezy_ppm_fuzz"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", str(ROOT / "include"),
                    str(ROOT / "tests" / "fuzz" / "ppm_fuzz.cpp"), "-o", str(exe), "-L", str(ROOT / "jpezy_amd"), "-ljpezy_hip",
                    "-Wl,-rpath," + str(ROOT / "jpezy_amd"), "-pthread"], check=True)
    rng = np.random.default_rng(1)

    def ppm(w, h, extra=""):
        px = rng.integers(0, 256, (w * h, 3))
        return ("P3\n" + extra + f"{w} {h}\n255\n" + "\n".join(" ".join(map(str, q)) for q in px) + "\n").encode()

    seeds_ = [ppm(4, 3), ppm(16, 16, "# c\n"), ppm(1, 1), ppm(33, 7), b"P3\n2 2\n255\n1 2 3\n4 5 6\n7 8 9\n10 11 12"]
    alphabet = b"0123456789 \n\t#-+P3\r\x00\xffa"
    with tempfile.TemporaryDirectory() as td:
        names = []
        for i in range(n):
            b = bytearray(seeds_[i % len(seeds_)])
            for _ in range(int(rng.integers(1, 8))):
                op, pos = int(rng.integers(0, 4)), int(rng.integers(0, max(1, len(b))))
                if op == 0 and b:
                    b[pos] = alphabet[int(rng.integers(0, len(alphabet)))]
                elif op == 1:
                    b[pos:pos] = bytes(alphabet[int(k)] for k in rng.integers(0, len(alphabet), int(rng.integers(1, 6))))
                elif op == 2:
                    del b[pos:pos + int(rng.integers(1, 6))]
                else:
                    del b[int(rng.integers(0, len(b) + 1)):]
            q = Path(td) / f"m{i}.ppm"
            q.write_bytes(bytes(b))
            names.append(str(q))
        lst = Path(td) / "l.txt"
        lst.write_text("\n".join(names) + "\n")
        r = subprocess.run([str(exe), "@" + str(lst)], capture_output=True, text=True)
    print("P3 reader:", r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "")
    if r.returncode != 0:
        print(r.stderr[-4000:])
    return r.returncode


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    exe = Path(tempfile.gettempdir()) / "jpezy_host_codec_fuzz"
    src = [ROOT / "tests" / "fuzz" / "host_codec_fuzz.cpp", ROOT / "jpezy_amd" / "csrc" / "jpezy_host_codec.cpp"]
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                    "-I", str(ROOT / "include"), *map(str, src), "-o", str(exe), "-pthread"], check=True)
    rng = np.random.default_rng(2026)
    ss = seeds()
    with tempfile.TemporaryDirectory() as td:
        names = []
        for i, s in enumerate(ss):
            p = Path(td) / f"seed{i}.jpg"
            p.write_bytes(s)
            names.append(str(p))
        for i in range(n):
            p = Path(td) / f"m{i}.jpg"
            p.write_bytes(mutate(ss[i % len(ss)], rng))
            names.append(str(p))
        lst = Path(td) / "list.txt"
        lst.write_text("\n".join(names) + "\n")
        r = subprocess.run([str(exe), "@" + str(lst)], capture_output=True, text=True)
        print(r.stdout.strip())
        if r.returncode != 0:
            print(r.stderr[-4000:])
            return r.returncode
        # the core of the GPU Huffman decoder (tables + decode step) walked on the CPU over the same files, same sanitizers
        exe2 = Path(tempfile.gettempdir()) / "jpezy_huffdec_core_fuzz"
        src2 = [ROOT / "tests" / "fuzz" / "huffdec_core_fuzz.cpp", ROOT / "jpezy_amd" / "csrc" / "jpezy_host_codec.cpp"]
        subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                        "-I", str(ROOT / "include"), *map(str, src2), "-o", str(exe2), "-pthread"], check=True)
        r = subprocess.run([str(exe2), "@" + str(lst)], capture_output=True, text=True)
    print(r.stdout.strip())
    if r.returncode != 0:
        print(r.stderr[-4000:])
        return r.returncode
    if (ROOT / "jpezy_amd" / "libjpezy_hip.so").exists():       # the reader's header pulls in the class surface, which links the library
        return ppm_phase(max(200, n // 4))
    return 0


if __name__ == "__main__":
    sys.exit(main())
