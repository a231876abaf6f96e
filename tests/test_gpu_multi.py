"""Multi-GPU entry points on the GPU: the in-process sharded calls (host and
device-resident), a cold kernel cache under the one-thread-per-context launch, the RCCL output
gather, and bench.py launching its own ranks.  The GPU box has ONE card: several contexts (or
ranks) share it where the test says so; the two-device cases skip themselves there."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import famseq_amd as fs
from _cases import load_cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BY = {c.name: c for c in load_cases()}


def _ped10_batch(n):
    c = BY["bn_synth:ped10"]
    reps = n // c.lk.shape[0] + 1
    return c, np.tile(c.lk, (reps, 1, 1))[:n].copy(), np.tile(c.flags, reps)[:n].copy()


def test_device_resident_sharded_entry():
    """famseq_bn_batch_device_sharded: every shard's arrays already on its context's device; one
    host thread per context; same bits as one famseq_bn_batch call over the whole batch."""
    import torch

    c, lk, flags = _ped10_batch(3001)
    model = fs.make_model(c.pedigree())
    one = fs.Context(model, enum_impl=1)  # one kernel on both sides: same bits
    ref = one.bn_batch(lk, flags)
    one.close()
    ctxs = [fs.Context(model, enum_impl=1) for _ in range(3)]
    dev = torch.device("cuda", 0)
    cuts = [fs.shard.site_range(len(lk), r, 3) for r in range(3)]
    d_lk = [torch.from_numpy(lk[a:b]).to(dev) for a, b in cuts]
    d_fl = [torch.from_numpy(flags[a:b]).to(dev) for a, b in cuts]
    d_post = [torch.empty_like(x) for x in d_lk]
    d_single = [torch.empty_like(x) for x in d_lk]
    d_st = [torch.empty(b - a, dtype=torch.uint8, device=dev) for a, b in cuts]
    torch.cuda.synchronize()
    p = lambda ts: [t.data_ptr() for t in ts]
    fs.bn_batch_device_sharded(ctxs, [b - a for a, b in cuts], p(d_lk), p(d_fl), p(d_post), p(d_single), p(d_st))
    got = [torch.cat(x).cpu().numpy() for x in (d_post, d_single, d_st)]
    for a, b in zip(got, ref):
        assert np.array_equal(a, b, equal_nan=True)
    # optional arrays absent, an empty shard in the middle
    d_post2 = [torch.zeros_like(x) for x in d_lk]
    fs.bn_batch_device_sharded(ctxs, [cuts[0][1] - cuts[0][0], 0, cuts[2][1] - cuts[2][0]], p(d_lk), None, p(d_post2))
    assert bool((d_post2[1] == 0).all())
    with pytest.raises(fs.FamseqError, match="no CPU path"):
        fs.bn_batch_device_sharded([fs.Context(model, device=-1)], [4], p(d_lk[:1]), None, p(d_post[:1]))
    for x in ctxs:
        x.close()


def test_sharded_threads_on_a_cold_kernel_cache(tmp_path, monkeypatch, capfd):
    """Three host threads reach the JIT together with an empty cache directory (scratch
    files named by hash + pid only let them clobber each other and latch the slow kernel).  Every
    context must end up on the generated lane kernel, with the right answer."""
    monkeypatch.setenv("FAMSEQ_KERNEL_CACHE", str(tmp_path))
    c, lk, flags = _ped10_batch(1800)
    model = fs.make_model(c.pedigree())
    ref_ctx = fs.Context(model, enum_impl=0)
    ref = ref_ctx.bn_batch(lk, flags)
    ref_ctx.close()
    assert not list(tmp_path.iterdir())
    ctxs = [fs.Context(model, lane_min_sites=1) for _ in range(3)]
    got = fs.bn_batch_sharded(ctxs, lk, flags)
    assert np.array_equal(got[2], ref[2]) and np.array_equal(got[1], ref[1], equal_nan=True)
    np.testing.assert_allclose(got[0], ref[0], rtol=1e-9, atol=0)
    for x in ctxs:
        plan = x.plan()
        objs = [plan["enum_lane_code_object"]] + plan["enum_group_code_objects"]
        assert plan["enum_lane_failed"] == 0 and any(o.startswith(str(tmp_path)) for o in objs), plan
        x.close()
    left = sorted(f.name for f in tmp_path.iterdir())
    assert all(f.endswith((".hsaco", ".res")) for f in left), left  # no stray sources, logs or temporaries
    assert "unavailable" not in capfd.readouterr().err


def test_lane_kernel_failure_is_reported(tmp_path, monkeypatch, capfd):
    """No compiler and nothing in the cache: auto mode falls back to the team kernel, says so on
    stderr once and in the plan; asking for the lane kernel explicitly is an error."""
    monkeypatch.setenv("FAMSEQ_KERNEL_CACHE", str(tmp_path))
    monkeypatch.setenv("FAMSEQ_HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("FAMSEQ_NO_HIPRTC", "1")
    c, lk, flags = _ped10_batch(300)
    model = fs.make_model(c.pedigree())
    ctx = fs.Context(model, lane_min_sites=1)
    post, single, st = ctx.bn_batch(lk, flags)
    ref = fs.Context(model, enum_impl=0)
    want = ref.bn_batch(lk, flags)
    ref.close()
    assert np.array_equal(post, want[0], equal_nan=True)
    assert ctx.plan()["enum_lane_failed"] == 1 and "compilation failed" in ctx.plan()["enum_lane_error"]
    ctx.bn_batch(lk, flags)
    err = capfd.readouterr().err
    assert err.count("enumeration kernel is unavailable") == 1, err
    with pytest.raises(fs.FamseqError, match="unavailable"):
        ctx.set_option("enum_impl", 1)
    ctx.close()


def test_jit_refuses_to_spawn_a_compiler_under_a_profiler(tmp_path, monkeypatch):
    """(Only a host without libhiprtc ever gets here: the in-process compiler needs no child process.)"""
    monkeypatch.setenv("FAMSEQ_KERNEL_CACHE", str(tmp_path))
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    monkeypatch.setenv("FAMSEQ_NO_HIPRTC", "1")
    monkeypatch.setenv("FAMSEQ_QUIET", "1")
    c, _, _ = _ped10_batch(1)
    ctx = fs.Context(fs.make_model(c.pedigree()), device=-1)
    with pytest.raises(fs.FamseqError, match="profiler is attached"):
        ctx.set_option("engine", fs.ENGINE_ELIM)
    ctx.close()


def test_jit_compiles_in_process_under_a_profiler(tmp_path, monkeypatch):
    """With libhiprtc the compile needs no child process, so a profiler's environment does not stop it (round 1
    ran hipcc through std::system from the profiled process: an exec from a GPU-initialised process)."""
    monkeypatch.setenv("FAMSEQ_KERNEL_CACHE", str(tmp_path))
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    monkeypatch.setenv("FAMSEQ_HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("FAMSEQ_QUIET", "1")
    c, _, _ = _ped10_batch(1)
    ctx = fs.Context(fs.make_model(c.pedigree()), device=-1)
    ctx.set_option("engine", fs.ENGINE_ELIM)
    assert ctx.plan()["elim_code_object"].startswith(str(tmp_path))
    ctx.close()


def _nccl_worker(rank, world, port, n_sites, out):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    ped = fs.synthetic_pedigree("ped5")
    mo, fa = ped.relations()
    lo, hi = fs.shard.site_range(n_sites, rank, world)
    lk, fl = fs.synth.gen_batch_torch(mo.tolist(), fa.tolist(), hi - lo, 1, first_site=lo, device=dev)
    post = torch.empty_like(lk)
    ctx = fs.Context(fs.make_model(ped), device=rank, enum_impl=1)  # one kernel on both sides: same bits
    ctx.bn_batch_device(hi - lo, lk.data_ptr(), fl.data_ptr(), post.data_ptr(), 0, 0, torch.cuda.current_stream().cuda_stream)
    full = fs.shard.gather_sites(post, n_sites)  # RCCL all_gather on the device tensors
    t = fs.shard.max_over_ranks(1.0 + rank, device=dev)
    if rank == 0:
        assert t == float(world)
        np.save(out, full.cpu().numpy())
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


def _run_gather(world, tmp_path):
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "full.npy")
    n_sites = 20001
    mp.spawn(_nccl_worker, args=(world, port, n_sites, out), nprocs=world, join=True)
    ped = fs.synthetic_pedigree("ped5")
    mo, fa = ped.relations()
    lk, fl = fs.synth.gen_batch(mo, fa, n_sites, 1)
    ctx = fs.Context(fs.make_model(ped), enum_impl=1)
    ref = ctx.bn_batch(lk, fl)[0]
    ctx.close()
    assert np.array_equal(np.load(out), ref)


def test_rccl_gather_one_rank(tmp_path):
    """shard.gather_sites through RCCL (backend "nccl") with the device tensors of a real run: the
    one-rank group this box can form (a second rank on the same card is refused by RCCL)."""
    _run_gather(1, tmp_path)


def test_rccl_gather_two_ranks(tmp_path):
    if fs.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's 8-GPU node runs it)")
    _run_gather(2, tmp_path)


def _bench(args, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]  # stdout is the ONE JSON line, nothing else
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_launches_its_own_ranks(scaling):
    """`python bench.py --gpus 2` as a plain command (the driver's SCALE shape): bench.py starts the
    ranks itself before touching the GPU.  On this 1-GPU box the two ranks share the card."""
    share = [] if fs.device_count() >= 2 else ["--share-gpu"]
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "ped5", "--sites", "200000",
                  "--scaling", scaling, "--no-cpu-baseline"] + share)
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["value"] > 0
    cfg = out["config"]
    assert cfg["global_sites"] == (400000 if scaling == "weak" else 200000)
    assert cfg["barrier_backend"].startswith("gloo" if share else "nccl")
    k = out["per_rank_kernel_ms"]
    assert len(k["all"]) == 2 and 0 < k["min"] <= k["max"]
    assert out["roofline"]["frac"] > 0


def test_multi_rank_bench_line_carries_the_strong_scaling_case():
    """For N > 1 the headline (weak scaling, BASELINE configs[2] per GPU) comes with `configs_3_strong`: configs[3] read
    literally — 10 M ten-member sites in total, sharded — timed by the same ranks; plus every rank's verdict on its outputs and
    the world size the backend reports.  (Here: two ranks sharing the one GPU, 300 k resident sites each, so each rank's share
    of the 10 M is capped by what it holds.)"""
    share = [] if fs.device_count() >= 2 else ["--share-gpu"]
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--sites", "300000", "--no-cpu-baseline", "--no-elim"] + share)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["per_rank_outputs_valid"] == [True, True]
    assert out["config"]["world_size_reported_by_backend"] == 2
    st = out["configs_3_strong"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 2 and st["value"] > 0 and st["per_rank_outputs_valid"] == [True, True]
    assert st["sites_per_gpu"] == 300000 and st["global_sites"] == 600000


def test_tune_times_the_variants_and_later_contexts_start_from_the_pick(tmp_path, monkeypatch):
    """famseq_set_option "tune": the candidates of a pedigree's generated kernels are timed on this GPU and the winners'
    indices kept as notes in the kernel cache; results stay what they were; a later context compiles the picked variant
    only.  (A twelve-member, three-family pedigree: both block shapes exist, both fence variants apply.)"""
    import json

    monkeypatch.setenv("FAMSEQ_KERNEL_CACHE", str(tmp_path))
    ped = fs.synthetic_pedigree("ped15:12")
    model = fs.make_model(ped)
    mo, fa = ped.relations()
    lk, flags = fs.synth.gen_batch(mo, fa, 300, 4)
    ctx = fs.Context(model, enum_impl=1)
    before = ctx.bn_batch(lk, flags)
    ctx.set_option("tune", 1)
    plan = ctx.plan()
    assert "enumeration (7- / 6-member block): v0" in plan["tune"] and "sum-product (likelihoods re-read from LDS" in plan["tune"], plan["tune"]
    assert plan["enum_lane_variant"] in (0, 2) and ("loaded: enumeration v%d" % plan["enum_lane_variant"]) in plan["tune"]
    after = ctx.bn_batch(lk, flags)
    np.testing.assert_allclose(after[0], before[0], rtol=1e-12, atol=0)
    assert np.array_equal(after[1].view(np.uint64), before[1].view(np.uint64)) and np.array_equal(after[2], before[2])
    ctx.set_option("engine", fs.ENGINE_ELIM)
    by_elim = ctx.bn_batch(lk, flags)
    np.testing.assert_allclose(by_elim[0], before[0], rtol=1e-9, atol=0)
    picks = sorted(f.name for f in tmp_path.iterdir() if f.name.endswith(".pick"))
    assert len(picks) == 2
    lane_pick, elim_pick = plan["enum_lane_variant"], ctx.plan()["elim_variant"]
    assert elim_pick in (0, 1, 4, 5) and ("-> v%d" % elim_pick) in plan["tune"]  # the report names what gets loaded
    ctx.close()
    # a later context: same picks, nothing timed
    ctx = fs.Context(model, enum_impl=1)
    ctx.set_option("engine", fs.ENGINE_ELIM)
    p2 = ctx.plan()
    ctx.close()
    assert p2["enum_lane_variant"] == lane_pick and p2["elim_variant"] == elim_pick and p2["tune"] == ""

