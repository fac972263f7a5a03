"""The driver parses the LAST stdout line of bench.py and keeps only a tail of the output (round 3's 24 KB line arrived cut and did not
parse). The line bench.py prints is built by bench.compact_line(): strict JSON, < 4096 bytes, carrying the contract's keys plus a compact
`roofline` and `cpu_baseline`; everything else goes to bench_detail.json. CPU tier: built from recorded results, no GPU."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _reject(name):
    raise ValueError(f"non-strict JSON constant {name}")


RECORDED = [os.path.join(ROOT, "profiles", f) for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if f.endswith(".json") and "bench" in f]


def _load(path):
    txt = open(path).read().strip().splitlines()[-1]
    return json.loads(txt)


@pytest.mark.parametrize("path", RECORDED, ids=[os.path.basename(p) for p in RECORDED])
def test_compact_line_of_recorded_results(path):
    import bench
    d = _load(path)
    if "metric" not in d or "roofline" not in d:
        pytest.skip("not a bench line")
    line = bench.compact_line(d)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 4096, len(line)
    c = json.loads(line, parse_constant=_reject)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in c, k
    assert c["value"] == d["value"] and c["config"]["workload_id"] == d["config"]["workload_id"] and isinstance(c["config"]["workload"], str)
    r = c["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - d["roofline"]["frac"]) <= 1e-4 * abs(d["roofline"]["frac"]) + 1e-12
    if "cpu_baseline" in d:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c["cpu_baseline"], k
    for name, w in (d.get("workloads") or {}).items():
        assert abs(c["workloads"][name]["value"] - w["value"]) <= 1e-4 * w["value"]


def test_compact_line_survives_a_hostile_result():
    """NaN / inf never reach the line, arbitrarily long strings and many sub-workloads cannot push it over the limit."""
    import bench
    d = _load(os.path.join(ROOT, "profiles", "r03_bench_n1.json"))
    d["roofline"]["frac"] = float("nan")
    d["roofline"]["traffic"] = float("inf")
    d["config"]["workload"] = "x" * 50_000
    d["config"]["parallelism"] = "y" * 50_000
    d["cpu_baseline"]["sample"] = "z" * 50_000
    for i in range(200):
        d["workloads"][f"W{i}"] = dict(d["workloads"]["D"])
    line = bench.compact_line(d)
    assert len(line) < 4096
    c = json.loads(line, parse_constant=_reject)
    assert c["value"] == d["value"] and c["roofline"].get("frac") is None and c["roofline"]["traffic"] is None


def test_gpus_n_never_reports_another_n_with_exit_code_zero():
    """Round 5: `python bench.py --gpus N` (N > 1) without a launcher starts its own N ranks; with fewer than N visible devices — here: none — it
    prints ONE JSON error line and exits non-zero instead of quietly running the one-GPU workload as before. A launcher that started another number
    of ranks than --gpus is refused the same way."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.device_count() < 8:            # (on an 8-GPU box this command IS the full 8-rank benchmark: not a unit test's business)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 2, (r.returncode, r.stdout[-500:], r.stderr[-500:])
        d = json.loads(r.stdout.strip().splitlines()[-1], parse_constant=_reject)
        assert "error" in d and d["n_gpus_requested"] == 8 and d["devices_visible"] == torch.cuda.device_count() and "value" not in d
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)          # a launcher with another rank count
    assert r.returncode == 2
    d = json.loads(r.stdout.strip().splitlines()[-1], parse_constant=_reject)
    assert "error" in d and "value" not in d


def test_cached_inputs_are_verified_and_regenerated_on_mismatch(tmp_path):
    """.bench_cache/*.npz travels to the GPU box with the push: every file carries its generator parameters and a SHA-256 of its arrays;
    a file that does not match either is ignored and the inputs are generated again (same hash as a fresh generation)."""
    import numpy as np
    import bench
    cache = str(tmp_path)
    a = bench.make_inputs_nclt(0, cache_dir=cache)
    path = os.path.join(cache, "ctgn_bench_C_v2_r0.npz")
    assert os.path.exists(path)
    sha = bench.arrays_sha256(a)
    d = dict(np.load(path))
    assert str(d["__sha256__"]) == sha and "generator" in json.loads(str(d["__params__"]))
    assert bench.arrays_sha256(bench.make_inputs_nclt(0, cache_dir=cache)) == sha            # a clean load
    d["raw"] = d["raw"] + 1e-9                                                              # tampered arrays under the old hash
    np.savez(path, **d)
    assert bench.cache_load(path, json.loads(str(d["__params__"]))) is None
    assert bench.arrays_sha256(bench.make_inputs_nclt(0, cache_dir=cache)) == sha            # regenerated, not trusted
    d = dict(np.load(path))
    d["__params__"] = np.array(json.dumps({"workload": "something else"}))                   # another generator's file under this name
    np.savez(path, **d)
    assert bench.arrays_sha256(bench.make_inputs_nclt(0, cache_dir=cache)) == sha
    np.savez(path, raw=a["raw"])                                                             # a file from before the provenance fields
    assert bench.arrays_sha256(bench.make_inputs_nclt(0, cache_dir=cache)) == sha
