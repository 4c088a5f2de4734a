"""bench.py's bookkeeping (CPU): the algorithmic byte counts of SURVEY.md 8(d) / BASELINE.md, the workloads it names,
and the cpu_baseline leg (a bounded run of the oracle)."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_match_the_survey():
    b = _bench()
    # config A: 4096x4096 colour: 50.3 MB read + 50.3 MB written = 6 B/px
    assert b.algorithmic_bytes(4096, 4096, False, "encode") == 100663296 == 6 * 4096 * 4096
    assert b.algorithmic_bytes(4096, 4096, False, "decode") == 100663296
    # config C: 7680x4320 gray: 99.5 MB read + 66.4 MB written = 5 B/px
    assert b.algorithmic_bytes(7680, 4320, True, "encode") == 3 * 7680 * 4320 + 480 * 270 * 4 * 128 == 165888000
    # config B: 1920x1080, height padded to 1088 by edge replication: 6.22 MB read + 6.27 MB written per frame
    assert b.algorithmic_bytes(1920, 1080, False, "encode") == 3 * 1920 * 1080 + 120 * 68 * 6 * 128 == 6220800 + 6266880
    assert b.HBM_PEAK_GBS == 8000.0
    assert b.WORKLOADS["encode4096"][:5] == (4096, 4096, False, 1, "encode")
    assert set(b.WORKLOADS) == {"encode4096", "decode4096", "gray8k", "batch1080p", "gray8k_decode"}


def test_cpu_baseline_leg_runs_the_oracle_on_a_bounded_sample():
    b = _bench()
    r = b.cpu_baseline(256, 128, False, budget_s=0.3)
    assert r["unit"] == "Mpixels/s" and r["cores"] == 1 and r["kind"] == "port" and r["value"] > 1.0
    assert "MCU rows" in r["sample"]
