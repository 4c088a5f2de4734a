"""The unpinned constants (cos table, 1/sqrt 2, pad-bit polarity: SURVEY H1/H8) live in ONE header,
include/jpezy_constants.h, and both the oracle and the product follow it: built against an alternative header
(tools/gen/gen_constants.py --variant alt1: cosines one ULP larger, 1/sqrt 2 one ULP larger, pad bits 1) the two still agree
with each other bit for bit -- and disagree with the frozen build, so the test has teeth.  If the true SrookCppLibraries
values are ever obtained, regenerating that header is the whole change."""
import json
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


@pytest.fixture(scope="module")
def alt():
    from tools.gen import build_alt
    lib, ora, hdr = build_alt.build_alt("alt1")     # a no-op when __graft_entry__.build() already made them
    return lib, ora, hdr


def _probe(lib=None, ora=None, gpu=False):
    env = dict(os.environ)
    env.pop("JPEZY_LIB", None)
    env.pop("JPEZY_ORACLE_LIB", None)
    if lib:
        env["JPEZY_LIB"] = str(lib)
    if ora:
        env["JPEZY_ORACLE_LIB"] = str(ora)
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "_constants_probe.py")] + (["--gpu"] if gpu else []),
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_alt_header_changes_only_the_unpinned_constants(alt):
    frozen = (ROOT / "include" / "jpezy_constants.h").read_text().splitlines()
    other = Path(alt[2]).read_text().splitlines()
    assert len(frozen) == len(other)
    changed = [(a, b) for a, b in zip(frozen, other) if a != b]
    kinds = set()
    for a, b in changed:
        if a.startswith("/* GENERATED"):
            continue
        if "JPEZY_INV_SQRT2" in a:
            kinds.add("sqrt")
        elif "JPEZY_PAD_BIT" in a:
            kinds.add("pad")
            assert a.split()[-1] == "0" and b.split()[-1] == "1"
        else:
            assert re.match(r"^\s+-?0x1\.", a), a            # a row of the cos table
            kinds.add("cos")
    assert kinds == {"sqrt", "pad", "cos"}


def test_product_host_path_follows_the_constants_header(alt):
    base = _probe()
    other = _probe(lib=alt[0], ora=alt[1])
    assert base["all_equal"], base["verdict"]
    assert other["all_equal"], other["verdict"]              # alt product == alt oracle
    assert base["constants"] != other["constants"]
    # ... and the alternative constants really change results: coefficients of the flat-level frame (DC boundaries) and
    # every .jpg (pad bits at least)
    assert base["flat_coeffs"] != other["flat_coeffs"] and base["rand_coeffs"] != other["rand_coeffs"]
    assert all(base[k] != other[k] for k in base if k.endswith("_jpg"))
    # mixing the builds is detected (the product's tables are built from ITS header, not from the oracle's)
    mixed = _probe(lib=alt[0], ora=None)
    assert not mixed["all_equal"]


@pytest.mark.gpu
def test_gpu_kernels_follow_the_constants_header(alt):
    """device tables (ks, exact DC table, c_cos of the exact paths, dequantiser constants) are host-built from the header:
    the alt product's kernels and GPU entropy stage agree with the alt oracle, all force_exact levels included"""
    other = _probe(lib=alt[0], ora=alt[1], gpu=True)
    assert other["all_equal"], {k: v for k, v in other["verdict"].items() if not v}
    base = _probe(gpu=True)
    assert base["all_equal"], {k: v for k, v in base["verdict"].items() if not v}
    assert base["flat_coeffs"] != other["flat_coeffs"]
