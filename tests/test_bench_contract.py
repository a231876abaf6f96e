"""Static checks of the driver contract that do not need a GPU: bench.py's flags and helper
arithmetic, __graft_entry__'s entry points, the committed profile evidence."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_flags_and_helpers(monkeypatch):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.workload, a.engine) == (1, "ped10", "enum") and a.steps >= 1 and a.warmup >= 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1", "--workload", "ped15"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup, a.workload) == (8, 3, 1, "ped15")
    assert bench.WORKLOADS["ped10"] == (2, 10_000_000) and bench.WORKLOADS["ped5"] == (1, 1_000_000)
    assert bench.fp64_ops_per_site({}, 10) == 59049
    assert 1 <= bench.host_cores() <= (os.cpu_count() or 1)
    assert bench.HBM_PEAK_GBPS == 8000.0


def test_cpu_baseline_leg_runs_on_a_tiny_budget():
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    import famseq_amd as fs

    out = bench.cpu_baseline(fs.synthetic_pedigree("ped5"), 1, 0.3, 5)
    assert out["kind"] == "port" and out["unit"] == "sites/s" and out["value"] > 0 and out["cores"] >= 1


def test_graft_entry_points_exist():
    sys.path.insert(0, ROOT)
    g = importlib.import_module("__graft_entry__")
    assert callable(g.build) and callable(g.smoke)


def test_profile_evidence_is_committed():
    prof = os.path.join(ROOT, "profiles")
    rounds = sorted(d for d in os.listdir(prof) if os.path.isdir(os.path.join(prof, d)))
    assert rounds, "no rocprofv3 summaries under profiles/"
    latest = os.path.join(prof, rounds[-1])
    assert any(f.startswith("kernel_stats") for f in os.listdir(latest))
    t = json.load(open(os.path.join(prof, "hbm_traffic_ped10.json")))
    assert t["sites_per_launch"] > 0 and t["bytes_per_launch"] >= t["algorithmic_bytes_per_launch"] * 0.9
