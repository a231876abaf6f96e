"""Small batches of wide pedigrees: the lanes-per-site mode of the generated
enumeration kernel — 3^d lanes share a site, each enumerating one combination of the d outermost
looped members' digits, partial marginals summed through LDS by the group's first lane — against the
fixtures and the oracle, for every d, at batch sizes around its chunk boundaries; and the automatic
choice of d by batch size."""
import numpy as np
import pytest

import famseq_amd as fs
import oracle
from _cases import load_cases

pytestmark = pytest.mark.gpu
BY = {c.name: c for c in load_cases()}
RTOL = 1e-9


def check(got, ref, what):
    post, single, st = got
    assert np.array_equal(st, ref[2]), what
    ok, s_ok = (ref[2] & 3) == 0, (ref[2] & 3) != 1
    assert np.array_equal(single[s_ok].view(np.uint64), ref[1][s_ok].view(np.uint64)), what
    np.testing.assert_allclose(post[ok], ref[0][ok], rtol=RTOL, atol=0, err_msg=what)
    assert np.all(np.isnan(post[~ok])), what


@pytest.mark.parametrize("name", ["bn_synth:ped10", "bn_synth:ped10_x", "bn_synth:chain7", "bn_vcf:fam01", "bn_lk:fam01"])
def test_every_group_size_matches_the_fixtures(name):
    c = BY[name]
    model = fs.make_model(c.pedigree(), **c.consts)
    probe = fs.Context(model, device=-1)
    dmax = probe.plan()["enum_group_digits_max"]
    probe.close()
    assert dmax >= 1
    for d in range(1, dmax + 1):
        ctx = fs.Context(model, enum_impl=1, group_digits=d)
        check(ctx.bn_batch(c.lk, c.flags), (c.post, c.single, c.status), "%s d=%d" % (name, d))
        assert ctx.plan()["enum_group_digits_last"] == d
        ctx.close()


@pytest.mark.parametrize("d", [1, 2, 3, 4])
def test_batch_sizes_around_the_chunk_boundaries(d):
    """256-thread workgroups hold 256 // 3^d sites per chunk (85, 28, 9, 3) with idle lanes left over;
    batches of one site, one chunk +- 1, several chunks and a ragged tail, chrX and Known mixed in."""
    ped = fs.synthetic_pedigree("ped10")
    mo, fa = ped.relations()
    spc = 256 // 3 ** d
    lk, flags = fs.synth.gen_batch(mo, fa, 7 * spc + 2, 2)
    flags[::3] |= fs.FLAG_CHRX
    lk[5, 2] = [1.0, 1e-40, 1e-40]  # sharp likelihoods on one site do not make a shortcut site by themselves ...
    lk[6] = np.where(np.arange(3) == 0, 1.0, 1e-40)  # ... on every member they do (status 0x80)
    ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders).bn_batch(lk, flags, threads=8)
    assert ref[2][6] == 0x80
    ctx = fs.Context(fs.make_model(ped), enum_impl=1, group_digits=d)
    for n in (1, spc - 1, spc, spc + 1, 7 * spc + 2):
        check(ctx.bn_batch(lk[:n], flags[:n]), tuple(x[:n] for x in ref), "d=%d n=%d" % (d, n))
    # optional outputs absent
    post, single, st = ctx.bn_batch(lk[:spc + 1], flags[:spc + 1], want_single=False, want_status=False)
    np.testing.assert_allclose(post, ref[0][:spc + 1], rtol=RTOL)
    ctx.close()


def test_group_size_follows_the_batch_size():
    """Auto mode: the fewer sites, the more lanes per site; a batch that fills the chip by itself
    runs one lane per site.  Same answers whichever d served the call (to rounding)."""
    ped = fs.synthetic_pedigree("ped10")
    mo, fa = ped.relations()
    lk, flags = fs.synth.gen_batch(mo, fa, 140_000, 2)
    ctx = fs.Context(fs.make_model(ped), lane_min_sites=1, chunk_sites=1 << 20)
    seen = {}
    ref = None
    for n in (140_000, 40_000, 12_000, 4_000, 1_200, 30):
        post, single, st = ctx.bn_batch(lk[:n], flags[:n])
        seen[n] = ctx.plan()["enum_group_digits_last"]
        if ref is None:
            ref = post
        np.testing.assert_allclose(post, ref[:n], rtol=1e-12, atol=0)
        assert np.all(st == 0)
    assert seen[140_000] == 0 and seen[30] == 4 and seen[12_000] >= 1, seen
    assert all(seen[a] <= seen[b] for a, b in zip(list(seen)[:-1], list(seen)[1:])), seen
    with pytest.raises(fs.FamseqError, match="group_digits"):
        ctx.set_option("group_digits", 5)
    ctx.close()
    # a narrow pedigree has nothing to spread: its whole enumeration is one unrolled block
    trio = fs.Context(fs.make_model(fs.synthetic_pedigree("trio")))
    assert trio.plan()["enum_group_digits_max"] == 0
    trio.close()


def test_a_tiny_batch_takes_the_generated_kernel_only_when_nothing_has_to_be_compiled(tmp_path, monkeypatch):
    """Below "lane_min_sites" (256) the compiled-in team kernel used to answer every call, so that a tiny call never waits
    for a per-pedigree compile.  Since round 3 the generated kernel serves such a call too when its code object is loaded
    or on disk already (the pre-built pedigrees; anything this user has run before) — and still not otherwise."""
    ped = fs.synthetic_pedigree("ped10")
    mo, fa = ped.relations()
    lk, flags = fs.synth.gen_batch(mo, fa, 40, 2)
    ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders).bn_batch(lk, flags)
    ctx = fs.Context(fs.make_model(ped))  # defaults; the in-tree cache holds ped10's kernels (build())
    check(ctx.bn_batch(lk, flags), ref, "cached")
    assert ctx.plan()["enum_group_digits_last"] == 4  # 81 lanes per site served it
    ctx.close()
    monkeypatch.setenv("FAMSEQ_KERNEL_CACHE", str(tmp_path))  # an empty cache: the same call must not compile anything
    ctx = fs.Context(fs.make_model(ped))
    check(ctx.bn_batch(lk, flags), ref, "uncached")
    assert ctx.plan()["enum_group_digits_last"] == 0 and not any(f.name.endswith(".hsaco") for f in tmp_path.iterdir())
    ctx.close()
