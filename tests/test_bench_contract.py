"""bench.py's bookkeeping (CPU): the algorithmic byte counts of SURVEY.md 8(d) / BASELINE.md, the workloads it names,
the cpu_baseline leg (a bounded run of the oracle, encode and decode side, one thread and all cores), the rank launcher
of `--gpus N`, and profiles/traffic.json against the rocprofv3 summaries it cites."""
import importlib.util
import json
import os
import subprocess
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
    # gray decode: the kernel skips the chroma blocks (r = g = b = clamp(Y)): 2 B/px of luma coefficients + 3 B/px written = 5 B/px
    assert b.algorithmic_bytes(7680, 4320, True, "decode") == 480 * 270 * 4 * 128 + 3 * 7680 * 4320 == 165888000
    assert b.HBM_PEAK_GBS == 8000.0
    assert b.WORKLOADS["encode4096"][:5] == (4096, 4096, False, 1, "encode")
    assert set(b.WORKLOADS) == {"encode4096", "decode4096", "gray8k", "batch1080p", "gray8k_decode", "encode4096_jpg", "decode4096_jpg"}
    assert (b.BATCH_W, b.BATCH_H, b.BATCH_FRAMES) == (1920, 1080, 4096)          # BASELINE configs[3]


def test_cpu_baseline_times_the_stage_the_workload_replaces():
    b = _bench()
    enc = b.cpu_baseline(256, 128, False, "encode", budget_s=0.3)
    dec = b.cpu_baseline(256, 128, False, "decode", budget_s=0.3)
    for r in (enc, dec):
        assert r["unit"] == "Mpixels/s" and r["cores"] == 1 and r["kind"] == "port" and r["value"] > 1.0
        assert "MCU rows" in r["sample"]
        assert r["all_cores"]["cores"] >= 1 and r["all_cores"]["value"] > 1.0        # the all-cores leg, core count stated
        assert r["huffman_stage"]["value"] > 0 and r["total_1core"]["value"] < r["value"]
    assert "FDCT" in enc["stage"] and "jpezy_encoder.hpp" in enc["stage"]
    assert "IDCT" in dec["stage"] and "jpezy_decoder.hpp" in dec["stage"]         # decode workloads time the DECODE oracle
    assert "decode_huffman" in dec["huffman_stage"]["what"]
    gray = b.cpu_baseline(256, 128, True, "encode", budget_s=0.2)
    assert gray["value"] > 1.0 and gray["huffman_stage"]["value"] > 0


def test_spawn_ranks_starts_one_process_per_rank_with_the_torchrun_environment(tmp_path):
    b = _bench()
    prog = ("import os,sys; open(os.path.join(sys.argv[1], os.environ['RANK']), 'w').write("
            "' '.join(os.environ[k] for k in ('RANK','LOCAL_RANK','WORLD_SIZE','MASTER_ADDR','MASTER_PORT','HSA_ENABLE_IPC_MODE_LEGACY')))")
    assert b.spawn_ranks(3, [sys.executable, "-c", prog, str(tmp_path)]) == 0
    got = [(tmp_path / str(r)).read_text().split() for r in range(3)]
    assert [g[:4] for g in got] == [[str(r), str(r), "3", "127.0.0.1"] for r in range(3)]
    assert len({g[4] for g in got}) == 1 and got[0][5] == "0"
    # a failing rank fails the job and stops the others
    bad = "import os,sys,time; sys.exit(7) if os.environ['RANK']=='1' else time.sleep(30)"
    assert b.spawn_ranks(2, [sys.executable, "-c", bad], timeout=20) == 7


def test_gpus_n_without_a_launcher_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE: two ranks are started (here each stops at 'needs a HIP device' or,
    on a box with one GPU, at the rank/GPU check); with WORLD_SIZE set the process is a rank itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 2
    else:
        assert p.returncode != 0
        assert "rank 0 exited" in p.stderr or "rank 1 exited" in p.stderr
    env["WORLD_SIZE"] = "3"
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE=3" in p.stderr


def test_traffic_json_is_what_the_cited_summaries_say():
    spec = importlib.util.spec_from_file_location("make_traffic_json", ROOT / "tools" / "gen" / "make_traffic_json.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    have = json.loads((ROOT / "profiles" / "traffic.json").read_text())
    want = m.build()
    assert have == want
    for wl, e in have.items():
        if wl.startswith("_"):
            continue
        assert (ROOT / e["source"]).exists()
        assert e["bytes_per_launch"] == int(round((2 * e["fetch_size_kib"] + e["write_size_kib"]) * 1024))
    # the counters agree with the algorithmic bytes to within 2 % (no wasted re-reads): encode and decode 4096^2
    b = _bench()
    assert abs(have["encode4096"]["bytes_per_launch"] / b.algorithmic_bytes(4096, 4096, False, "encode") - 1) < 0.02
    # decode: the coefficient reads are exact; the non-temporal half-line plane stores are COUNTED at ~1.3x (tools/gen/make_traffic_json.py)
    dec = have["decode4096"]
    assert abs(2 * dec["fetch_size_kib"] * 1024 / (b.algorithmic_bytes(4096, 4096, False, "decode") / 2) - 1) < 0.02
    assert 0.98 < dec["write_size_kib"] * 1024 / (3 * 4096 * 4096) < 1.40
    # a roofline fraction's numerator may not exceed what the launch moves: for EVERY workload the counters' traffic is at least
    # 0.98 x the algorithmic bytes bench.py charges it (gray8k_decode was charged 6 B/px, moved 5)
    for wl, e in have.items():
        if wl.startswith("_"):
            continue
        W, H, gray, fps, direction, _ = b.WORKLOADS[wl]
        alg = b.algorithmic_bytes(W, H, gray, direction) * fps
        assert e["bytes_per_launch"] >= 0.98 * alg, (wl, e["bytes_per_launch"], alg)
        assert e["bytes_per_launch"] <= 1.10 * alg, (wl, e["bytes_per_launch"], alg)         # ... and no wasted re-reads either


def test_other_workloads_and_native_multi_are_part_of_the_default_line():
    """the default command measures BASELINE configs[2] and [4] and the native multi-GPU entry after the
    timed region (objects of their own, never part of `value`); the run itself needs a GPU (tests/test_gpu_parity.py), here: the
    pieces exist, name real workloads, and can be switched off for profiling passes."""
    import inspect
    b = _bench()
    src = inspect.getsource(b.measure_other_workloads)
    for name in ("decode4096", "gray8k", "gray8k_decode"):
        assert name in b.WORKLOADS and f'"{name}"' in src
    assert set(b.OTHER_KERNELS) == {"encode", "decode"}
    assert "jpezy_encode_batch_multi" in inspect.getsource(b.measure_native_multi) or "encode_batch_multi" in inspect.getsource(b.measure_native_multi)
    args = b.parse_args(["--no-others", "--no-native-multi"])
    assert args.no_others and args.no_native_multi and not b.parse_args([]).no_others
    run = inspect.getsource(b.run_rank)
    assert 'out["other_workloads"] = others' in run and 'out["native_multi_gpu"] = native' in run
