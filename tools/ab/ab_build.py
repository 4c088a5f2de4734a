#!/usr/bin/env python3
"""A/B builds of libjpezy_hip.so for kernel experiments (development aid).

    python tools/ab/ab_build.py name1:-DFOO=1 name2:"-DFOO=2 -DBAR" ...

builds ab/libjpezy_<name>.so (same sources, extra hipcc flags); `tools/ab/ab_run.sh` then benches each of them on the
GPU box through JPEZY_LIB.  `ab/` is git-ignored (*.so) but travels with the gpurun snapshot.
"""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from jpezy_amd import _build as B  # noqa: E402


def build(spec):
    name, _, flags = spec.partition(":")
    out = ROOT / "ab" / name
    out.mkdir(parents=True, exist_ok=True)
    objs = []
    for src in B.LIB_SOURCES + B.LAB_SOURCES:          # A/B builds are laboratory builds: variants 2 / 3 and the probe switches are in
        obj = out / (src.stem + ".o")
        fl = B.COMMON + (B.DEVICE if src.suffix == ".hip" else ["-x", "c++"]) + B.LAB_FLAGS + flags.split()
        subprocess.run([B.HIPCC, *fl, "-c", str(src), "-o", str(obj)], check=True, capture_output=True)
        objs.append(str(obj))
    lib = ROOT / "ab" / f"libjpezy_{name}.so"
    subprocess.run([B.HIPCC, "-shared", "-fPIC", *B.DEVICE, "-o", str(lib), *objs], check=True, capture_output=True)
    for o in objs:
        Path(o).unlink()
    return lib


if __name__ == "__main__":
    with ThreadPoolExecutor(4) as ex:
        for lib in ex.map(build, sys.argv[1:]):
            print("built", lib)
