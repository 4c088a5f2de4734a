#!/usr/bin/env python3
"""Test infrastructure (tests/test_constants_override.py, __graft_entry__.build()): the product library AND the oracle built
against an alternative constants header (tools/gen/gen_constants.py --variant <variant>), to prove that both follow the one
header include/jpezy_constants.h.  Lives outside the product package because it builds the oracle.

    python tools/gen/build_alt.py [variant]
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from jpezy_amd import _build as B  # noqa: E402

ALT_DIR = B.PKG / "_alt"          # git-ignored *.so / *.h, regenerated here


def build_alt(variant="alt1", force=False):
    """Returns (product .so, oracle .so, header)."""
    hdr = ALT_DIR / f"jpezy_constants_{variant}.h"
    gen = ROOT / "tools" / "gen" / "gen_constants.py"
    if force or B._stale(hdr, [gen]):
        ALT_DIR.mkdir(parents=True, exist_ok=True)
        B._run([sys.executable, gen, "--variant", variant, "--out", hdr])
    lib = B.build_lib(force, constants=hdr, out=ALT_DIR / f"libjpezy_hip_{variant}.so")
    from oracle import oracle as O
    ora = O.build(force, constants=hdr, out=ROOT / "oracle" / "_alt" / f"libjpezy_oracle_{variant}.so")
    return lib, ora, hdr


if __name__ == "__main__":
    print(*build_alt(sys.argv[1] if len(sys.argv) > 1 else "alt1"), sep="\n")
