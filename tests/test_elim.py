"""Elimination engine, host side (no GPU): the per-pedigree kernel source is generated and
compiles for gfx950; pedigrees with loops are rejected (they stay on the enumeration engine)."""
import os

import pytest

import famseq_amd as fs


def cousins_marry():
    #  1,2 -> 3,4 ; 3x5 -> 7 ; 4x6 -> 8 ; 7x8 -> 9  (first-cousin marriage: a loop)
    ids = [1, 2, 3, 4, 5, 6, 7, 8, 9]
    mids = [0, 0, 2, 2, 0, 0, 5, 4, 8]
    fids = [0, 0, 1, 1, 0, 0, 3, 6, 7]
    gen = [1, 2, 1, 2, 2, 1, 1, 2, 1]
    return fs.Pedigree(ids, mids, fids, gen, ["s%d" % i for i in ids])


@pytest.mark.parametrize("name", ["ped5", "ped10"])
def test_generated_kernel_compiles(name):
    ctx = fs.Context(fs.make_model(fs.synthetic_pedigree(name)), device=-1)
    assert ctx.plan()["elim_supported"] == 1
    ctx.set_option("engine", fs.ENGINE_ELIM)
    p = ctx.plan()
    assert p["engine"] == fs.ENGINE_ELIM
    assert p["elim_code_object"].endswith(".hsaco") and os.path.getsize(p["elim_code_object"]) > 1000
    ctx.close()


def test_loop_pedigree_is_served_by_conditioning():
    """A first-cousin marriage closes a loop in the member/family graph: the engine conditions on
    one member (three passes of message passing per site) instead of refusing the pedigree."""
    ped = cousins_marry()
    ped.relations()
    ctx = fs.Context(fs.make_model(ped), device=-1)
    assert ctx.plan()["elim_supported"] == 1 and ctx.plan()["elim_conditioned_members"] == 1
    ctx.set_option("engine", fs.ENGINE_ELIM)
    assert ctx.plan()["engine"] == fs.ENGINE_ELIM
    with pytest.raises(fs.FamseqError):
        ctx.set_option("engine", 7)
    ctx.close()


def test_too_many_loops_stay_on_enumeration():
    """Five marriages between two sibships are four independent loops and need four conditioned
    members: beyond the engine's limit of three, so it says so and the ctx stays on enumeration."""
    ids, mids, fids, gen = [1, 2, 3, 4], [0, 0, 0, 0], [0, 0, 0, 0], [1, 2, 1, 2]
    sons, daughters = [], []
    for k in range(5):  # five sons of couple 1x2, five daughters of couple 3x4
        ids.append(len(ids) + 1); mids.append(2); fids.append(1); gen.append(1); sons.append(ids[-1])
        ids.append(len(ids) + 1); mids.append(4); fids.append(3); gen.append(2); daughters.append(ids[-1])
    for fa, mo in zip(sons, daughters):
        ids.append(len(ids) + 1); mids.append(mo); fids.append(fa); gen.append(1)
    assert len(ids) == 19
    ped = fs.Pedigree(ids, mids, fids, gen, ["s%d" % i for i in ids])
    ped.relations()
    ctx = fs.Context(fs.make_model(ped), device=-1)
    assert ctx.plan()["elim_supported"] == 0
    with pytest.raises(fs.FamseqError, match="conditioning"):
        ctx.set_option("engine", fs.ENGINE_ELIM)
    assert ctx.plan()["engine"] == fs.ENGINE_ENUM
    ctx.close()


def test_variant_is_the_first_that_does_not_spill(tmp_path, monkeypatch):
    """The generators emit variants from most to least instruction-level parallelism; the JIT takes
    the first one the compiler reports spill-free and leaves that report next to each code object.  With one-wave
    workgroups and no register cap nothing spills to scratch, and where to start is a measured rule
    (elim_first_variant: from five members on the family that keeps the likelihoods in registers, fence-free).
    (A fresh cache directory: every candidate is really compiled here.)"""
    monkeypatch.setenv("FAMSEQ_KERNEL_CACHE", str(tmp_path))
    picked = {}
    for name in ("ped5", "ped10"):
        ctx = fs.Context(fs.make_model(fs.synthetic_pedigree(name)), device=-1)
        ctx.set_option("engine", fs.ENGINE_ELIM)
        p = ctx.plan()
        ctx.close()
        picked[name] = p["elim_variant"]
        obj = p["elim_code_object"]
        assert obj.startswith(str(tmp_path))
        assert int(open(obj[:-6] + ".res").read()) == 0  # the variant in use has no scratch
    assert picked == {"ped5": 4, "ped10": 4}  # nothing spills, nothing has been measured here: registers-first, fence-free
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".res")]) == 2  # nothing else was compiled


def test_a_measured_pick_is_where_the_picker_starts(tmp_path, monkeypatch):
    """famseq_set_option "pick_elim" / "pick_lane" leave the note "tune" would (how build() ships its table of measured
    picks, famseq_amd/tuned_picks.json): every later context for the pedigree compiles that variant, and only it."""
    monkeypatch.setenv("FAMSEQ_KERNEL_CACHE", str(tmp_path))
    model = fs.make_model(fs.synthetic_pedigree("ped15"))
    ctx = fs.Context(model, device=-1)
    ctx.set_option("pick_elim", 1)
    ctx.set_option("pick_lane", 2)
    with pytest.raises(fs.FamseqError):
        ctx.set_option("pick_elim", 99)
    ctx.close()
    ctx = fs.Context(model, device=-1)
    ctx.set_option("enum_impl", 1)
    ctx.set_option("engine", fs.ENGINE_ELIM)
    p = ctx.plan()
    ctx.close()
    assert p["elim_variant"] == 1 and p["enum_lane_variant"] == 2
    assert "= 729 configurations per step" in p["enum_lane_shape"]  # the 6-member block
    assert sorted(f[-5:] for f in os.listdir(tmp_path)).count(".pick") == 2
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]) == 2


def test_the_shipped_pick_table_matches_the_pedigrees_build_prebuilds():
    import json
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge

    table = json.load(open(ge.PICKS))
    peds, _ = ge.build_pedigrees()
    keys = {ge.pedigree_key(p) for p in peds}
    # every shipped pick belongs to a pedigree whose kernels build() pre-builds (the table may lag behind the build set:
    # the fixtures' and the wide pedigrees joined it in round 3 and run on the static rules), the benchmark pedigrees are in it
    assert set(table) <= keys, (len(keys), len(table))
    import famseq_amd as fs_
    assert all(ge.pedigree_key(fs_.synthetic_pedigree(n)) in table for n in ("ped5", "ped10", "ped15", "trio", "quad"))
    assert all(v["lane"] in (0, 2) and v["elim"] in (-1, 0, 1, 4, 5) for v in table.values())


def test_a_spilling_variant_is_passed_over(tmp_path, monkeypatch):
    """The picker's own rule, on the form that does spill: 256-lane workgroups at two waves per SIMD (256 registers)
    from variant 0 — the ten-member fence-free variant is compiled, reports scratch, and loses to the next."""
    monkeypatch.setenv("FAMSEQ_KERNEL_CACHE", str(tmp_path))
    monkeypatch.setenv("FAMSEQ_ELIM_BT", "256")
    monkeypatch.setenv("FAMSEQ_ELIM_MINWAVES", "2")
    monkeypatch.setenv("FAMSEQ_VARIANT_MIN", "0")
    ctx = fs.Context(fs.make_model(fs.synthetic_pedigree("ped10")), device=-1)
    ctx.set_option("engine", fs.ENGINE_ELIM)
    p = ctx.plan()
    ctx.close()
    notes = sorted(int(open(os.path.join(tmp_path, f)).read()) for f in os.listdir(tmp_path) if f.endswith(".res"))
    assert p["elim_variant"] >= 1 and notes[0] == 0 and notes[-1] > 0
    assert len(notes) == p["elim_variant"] + 1
    # the loser's object stays too: another process sharing the cache may just have been handed its path
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]) == p["elim_variant"] + 1
