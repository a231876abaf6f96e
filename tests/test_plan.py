"""Host-side enumeration plan (famseq_amd/csrc/plan.cpp) checked on the CPU:
structure invariants for the benchmark pedigrees and a numpy replay of the plan's index
tables against the oracle (every 3^N configuration visited once, binned correctly)."""
import numpy as np
import pytest

import famseq_amd as fs
import oracle
from _cases import load_cases
from plan_emulator import emulate_site

CASES = {c.name: c for c in load_cases(("bn_vcf.npz", "bn_synth.npz"))}


def plan_for(ped, **opt):
    ctx = fs.Context(fs.make_model(ped), device=-1, **opt)
    p = ctx.plan()
    ctx.close()
    return p


@pytest.mark.parametrize("name,N,L,A,J,teams", [("ped5", 5, 3, 2, 0, 28), ("ped10", 10, 4, 4, 2, 3),
                                                ("ped15", 15, 5, 4, 6, 3)])
def test_benchmark_pedigree_plans(name, N, L, A, J, teams):
    p = plan_for(fs.synthetic_pedigree(name))
    assert (p["N"], p["L"], p["A"], p["J"]) == (N, L, A, J)
    assert p["team_lanes"] == 3 ** A and p["teams_per_block"] == teams
    assert p["L"] + p["A"] + p["J"] == N
    members = sorted(p["low_member"] + p["fixed_member"] + p["iter_member"])
    assert members == list(range(N))
    assert p["jn"][0] * p["jn"][1] * p["jn"][2] == 3 ** J
    assert p["lds_bytes"] <= 160 * 1024
    assert p["cols"] == 3 * L + 3 * J + 1


def test_low_members_are_childless_and_parents_are_high():
    for name in ("ped5", "ped10", "ped15"):
        ped = fs.synthetic_pedigree(name)
        mo, fa = ped.relations()
        p = plan_for(ped)
        low = set(p["low_member"])
        for i in range(ped.n):
            assert mo[i] not in low and fa[i] not in low


def test_options_change_the_tiling():
    ped = fs.synthetic_pedigree("ped10")
    p = plan_for(ped, fixed_digits=6)
    assert (p["A"], p["J"], p["block_threads"]) == (6, 0, 768)
    p = plan_for(ped, low_members=2)
    assert p["L"] == 2 and p["A"] + p["J"] == 8
    with pytest.raises(fs.FamseqError):
        plan_for(ped, block_threads=100)
    with pytest.raises(fs.FamseqError):
        plan_for(ped, no_such_option=1)


def test_max_size_pedigree_plans():
    """N = 20 (the reference CUDA limit): a 4-generation line with spouses."""
    ids, mids, fids, gen = [1, 2], [0, 0], [0, 0], [1, 2]
    while len(ids) < 20:
        child, spouse = len(ids) + 1, len(ids) + 2
        ids += [child, spouse]
        mids += [ids[-3] if gen[-1] == 2 else ids[-4], 0]
        fids += [ids[-4] if gen[-1] == 2 else ids[-3], 0]
        gen += [1, 2]
    ped = fs.Pedigree(ids[:20], mids[:20], fids[:20], gen[:20], ["s%d" % i for i in ids[:20]])
    ped.relations()
    p = plan_for(ped)
    assert p["N"] == 20 and p["L"] + p["A"] + p["J"] == 20 and p["jlevels"] <= 3


EMU = ["bn_synth:quad", "bn_synth:trio_lrc", "bn_synth:founders3", "bn_synth:ped5", "bn_synth:ped5_custom",
       "bn_synth:ped5_x", "bn_synth:chain7", "bn_vcf:fam04", "bn_vcf:fam06"]


@pytest.mark.parametrize("name", EMU)
@pytest.mark.parametrize("opt", [{}, {"fixed_digits": 1}, {"low_members": 1, "fixed_digits": 2}], ids=["auto", "A1", "L1A2"])
def test_plan_replay_matches_oracle(name, opt):
    c = CASES[name]
    ped = c.pedigree()
    model = fs.make_model(ped, **c.consts)
    ctx = fs.Context(model, device=-1, **opt)
    plan = ctx.plan()
    ctx.close()
    o = oracle.OracleModel(c.ids, c.mids, c.fids, c.genders, c.sequenced, **c.consts)
    post, _, st = o.bn_batch(c.lk, c.flags)
    full = np.nonzero(st == 0)[0][:3]
    assert len(full) > 0
    for s in full:
        bins, visited = emulate_site(plan, model, c.lk[s], int(c.flags[s]))
        assert visited == 3 ** c.n
        got = bins / bins.sum(axis=1, keepdims=True)
        np.testing.assert_allclose(got, post[s], rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("name", ["ped5", "ped10", "ped15"])
def test_generated_lane_kernel_accumulates_every_configuration_once(name, tmp_path, monkeypatch):
    """Structure of the generated enumeration source: (FMAs into the joint super-leaf accumulators
    in the unrolled block) x 3^(looped members) = 3^N, i.e. each joint configuration's weight is
    formed and added exactly once."""
    import re

    monkeypatch.setenv("FAMSEQ_KERNEL_CACHE", str(tmp_path))
    monkeypatch.setenv("FAMSEQ_KEEP_SRC", "1")
    ped = fs.synthetic_pedigree(name)
    ctx = fs.Context(fs.make_model(ped), device=-1)
    ctx.set_option("enum_impl", 1)
    p = ctx.plan()
    ctx.close()
    src = open(p["enum_lane_code_object"][:-6] + ".hip").read()
    leaf = len(re.findall(r"^\s*(S(?:_\d)+) = __builtin_fma\(\w+, W(?:_\w+)+, \1\);", src, flags=re.M))
    loops = len(re.findall(r"#pragma unroll 1\n\s*for \(int g\d+ = 0; g\d+ < 3; \+\+g\d+\)", src))
    m = re.search(r"(\d+) looped \+ (\d+) unrolled members", src.splitlines()[0])
    assert m and loops == int(m.group(1)) and int(m.group(1)) + int(m.group(2)) == ped.n
    assert leaf == 3 ** int(m.group(2))
    assert leaf * 3 ** loops == 3 ** ped.n
    # the accumulators are reduced into marginals exactly once, after all loops
    acc = set(re.findall(r"double (S(?:_\d)+) = 0;", src))
    assert len(acc) in (9, 27) and all(src.count(a + " = 0;") == 1 for a in acc)
