"""Builds the in-tree native artefacts with hipcc (gfx950 only; cross-compiles without a GPU).

    libjpezy_hip.so   the C-ABI library (include/jpezy_hip.h): HIP kernels + context + host Huffman/JFIF
    bin/jpezy_encode, bin/jpezy_decode   the CLIs (C++ host code linked against the library)
"""
import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
ROOT = PKG.parent
LIB = PKG / "libjpezy_hip.so"
BIN = PKG / "bin"

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -ffp-contract=off: the reference arithmetic is plain IEEE mul/add; every FMA in the kernels is explicit.
COMMON = ["-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
# -fno-slp-vectorize: hipcc otherwise pairs FP32 ops into v_pk_* (no faster on gfx950: tools/ubench/asm_rate.hip)
# and pays v_mov shuffles for it -- the f32 encode kernel is 10 % faster without (33.8 vs 37.5 us per 4096^2 frame)
DEVICE = ["--offload-arch=gfx950", "-fno-slp-vectorize"]

# The shipped library holds ONE f32 encode kernel (variant 1) beside the independent FP64 one (variant 0).  The persistent forms of
# round 5 (variants 2 / 3, jpezy_kernels_f32_ps.hip) and the timing probes (jpezy_lab.h) are the laboratory: `--lab` / with_lab=True /
# tools/ab/ab_build.py build them in (-DJPEZY_WITH_LAB); jpezy_ctx_set_variant(ctx, 2) says JPEZY_E_UNSUPPORTED otherwise.
LAB_SOURCES = [CSRC / "jpezy_kernels_f32_ps.hip"]
LAB_FLAGS = ["-DJPEZY_WITH_LAB"]
LIB_SOURCES = [CSRC / "jpezy_kernels.hip", CSRC / "jpezy_kernels_f32.hip", CSRC / "jpezy_kernels_generic.hip",
               CSRC / "jpezy_entropy.hip", CSRC / "jpezy_huffdec.hip",
               CSRC / "jpezy_capi.hip", CSRC / "jpezy_capi_entropy.hip", CSRC / "jpezy_capi_huffdec.hip", CSRC / "jpezy_capi_decode_batch.hip", CSRC / "jpezy_capi_multi.hip",
               CSRC / "jpezy_host_codec.cpp"]
LIB_DEPS = LIB_SOURCES + LAB_SOURCES + [CSRC / "jpezy_lab.h", CSRC / "jpezy_device.h", CSRC / "jpezy_f32_quad.h", CSRC / "jpezy_capi_internal.h", CSRC / "jpezy_experiment.h", CSRC / "jpezy_hostpipe.h", CSRC / "jpezy_host_codec.h", CSRC / "jpezy_entropy.h", CSRC / "jpezy_huffdec.h", CSRC / "jpezy_huffdec_core.h",
                          ROOT / "include" / "jpezy_hip.h", ROOT / "include" / "jpezy_constants.h"]
CLI = {"jpezy_encode": CSRC / "host" / "encode_main.cpp", "jpezy_decode": CSRC / "host" / "decode_main.cpp"}


def _stale(target, deps):
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.exists() and d.stat().st_mtime > t for d in deps)


def _run(cmd):
    proc = subprocess.run([str(c) for c in cmd], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("build failed:\n" + " ".join(map(str, cmd)) + "\n" + proc.stdout + proc.stderr)
    return proc.stdout + proc.stderr


def build_lib(force=False, verbose=False, constants=None, out=None, with_lab=False):
    """libjpezy_hip.so; constants/out: the same sources against an alternative constants header (force-included first,
    same include guard as include/jpezy_constants.h) into another file -- tests/test_constants_override.py.
    with_lab: the laboratory build (persistent encode variants 2 / 3 + probe switches), by default into libjpezy_hip_lab.so."""
    lib = Path(out) if out else (PKG / "libjpezy_hip_lab.so" if with_lab else LIB)
    deps = LIB_DEPS + ([Path(constants)] if constants else [])
    if force or _stale(lib, deps):
        objs = []
        extra = (["-include", str(Path(constants).resolve())] if constants else []) + (LAB_FLAGS if with_lab else [])
        objdir = CSRC if not out and not with_lab else lib.parent / (lib.stem + "_obj")
        objdir.mkdir(parents=True, exist_ok=True)
        for src in LIB_SOURCES + (LAB_SOURCES if with_lab else []):
            obj = objdir / (src.stem + ".o")
            flags = COMMON + (DEVICE if src.suffix == ".hip" else ["-x", "c++"]) + extra
            text = _run([HIPCC, *flags, "-c", src, "-o", obj])
            if verbose and text.strip():
                print(text)
            objs.append(obj)
        _run([HIPCC, "-shared", "-fPIC", *DEVICE, "-o", lib, *objs])
    return lib


def build_cli(force=False):
    BIN.mkdir(exist_ok=True)
    host_hdrs = list((CSRC / "host").glob("*.hpp"))
    for name, src in CLI.items():
        if not src.exists():
            continue
        exe = BIN / name
        if force or _stale(exe, [src, LIB, *host_hdrs]):
            _run([HIPCC, *COMMON, "-x", "c++", src, "-I", ROOT / "include", "-o", exe,
                  "-L", PKG, "-ljpezy_hip", "-Wl,-rpath,$ORIGIN/..", "-lpthread"])
    return BIN


def build_all(force=False, verbose=False):
    build_lib(force, verbose)
    build_cli(force)
    return LIB


if __name__ == "__main__":
    import sys
    if "--lab" in sys.argv:
        print("built", build_lib(force="--force" in sys.argv, verbose=True, with_lab=True))
    else:
        build_all(force="--force" in sys.argv, verbose=True)
        print("built", LIB)
