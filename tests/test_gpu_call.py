"""Fused call path (rows N2 + N4): packed integer PLs in, Phred-scaled posteriors and genotype
calls out, against the plain path plus the reference's host-side formulas."""
import numpy as np
import pytest

import famseq_amd as fs
from _cases import load_cases

pytestmark = pytest.mark.gpu
BY = {c.name: c for c in load_cases()}


def host_phred(p):
    """file.cpp:696-703: fabs(-10*log10(p)), +inf -> 99999."""
    with np.errstate(divide="ignore", invalid="ignore"):
        q = -10 * np.log10(p)
    return np.where(np.isposinf(q), 99999.0, np.abs(q))


@pytest.mark.parametrize("name", ["bn_vcf:fam01", "bn_vcf:fam04", "bn_synth:quad", "bn_synth:quad_mu0", "bn_synth:ped10_x", "bn_lk:fam06"])
def test_called_outputs_match_host_formulas(name):
    c = BY[name]
    ctx = fs.Context(fs.make_model(c.pedigree(), **c.consts))
    seq = np.nonzero(c.sequenced)[0][::-1].copy()  # a column order different from PED order
    post, single, st = ctx.bn_batch(c.lk, c.flags)
    gpp, fpp, fgt, st2 = ctx.bn_call_batch(seq, lk=c.lk, flags=c.flags)
    assert np.array_equal(st, st2)
    ok, s_ok = (st & 3) == 0, (st & 3) != 1
    np.testing.assert_allclose(gpp[s_ok], host_phred(single[s_ok][:, seq]), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(fpp[ok], host_phred(post[ok][:, seq]), rtol=1e-12, atol=1e-12)
    assert np.array_equal(fgt[ok], fs.call_genotypes(post[ok][:, seq]).reshape(-1, len(seq)))
    assert np.all(np.isnan(fpp[~ok])) and np.all(fgt[~ok] == -1) and np.all(np.isnan(gpp[~s_ok]))
    assert np.all(gpp[s_ok] >= 0) and np.all(fpp[ok] >= 0)
    ctx.close()


CALL_ENGINES = [("lane, fused", dict(enum_impl=1)), ("sum-product, fused", dict(engine=fs.ENGINE_ELIM)),
                ("team + separate stages", dict(enum_impl=0))]


@pytest.mark.parametrize("label,opt", CALL_ENGINES, ids=[e[0] for e in CALL_ENGINES])
@pytest.mark.parametrize("case", list(BY.values()), ids=list(BY))
def test_called_outputs_match_the_reference_posteriors(case, label, opt):
    """famseq_bn_call_batch against the COMPILED REFERENCE's posteriors (the fixtures: oracle/gen_golden.py), not against
    this library's own plain path: GPP / FPP = the drivers' fabs(-10 log10 p) (file.cpp:696-745) of the reference's single /
    BN posterior, FGT = its arg-max with ties to the lower genotype (family.cpp:636-665), status identical — on every
    fixture: TestData VCF sites and LK rows, chrX, custom priors, mu = 0, failed sites (status 1 and 2: NaN, -1), shortcut
    sites (FPP = GPP), for the fused call-path forms of both generated kernels and for the separate stages."""
    ped = case.pedigree()
    model = fs.make_model(ped, **case.consts)
    if "sum-product" in label:
        probe = fs.Context(model, device=-1)
        supported = probe.plan()["elim_supported"]
        probe.close()
        if not supported:
            pytest.skip("more loops than the sum-product engine conditions on")
    ctx = fs.Context(model, **opt)
    seq = np.nonzero(case.sequenced)[0][::-1].copy()
    gpp, fpp, fgt, st = ctx.bn_call_batch(seq, lk=case.lk, flags=case.flags)
    ctx.close()
    assert np.array_equal(st, case.status)
    ok, s_ok = (case.status & 3) == 0, (case.status & 3) != 1
    # posteriors agree with the reference to ~1e-15 relative; a Phred value q = -10 log10 p moves by 4.3 dp/p absolute,
    # so near p = 1 (q ~ 1e-10) the RELATIVE bar on q is loose and an absolute one of 1e-12 takes over
    np.testing.assert_allclose(gpp[s_ok], host_phred(case.single[s_ok][:, seq]), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(fpp[ok], host_phred(case.post[ok][:, seq]), rtol=1e-9, atol=1e-11)
    ref_call = fs.call_genotypes(case.post[ok][:, seq]).reshape(-1, len(seq))
    p_sorted = np.sort(case.post[ok][:, seq], axis=2)
    clear = p_sorted[:, :, 2] - p_sorted[:, :, 1] > 1e-12  # (a tie within rounding may fall either way)
    assert np.array_equal(fgt[ok][clear], ref_call[clear])
    assert np.all(np.isnan(fpp[~ok])) and np.all(fgt[~ok] == -1) and np.all(np.isnan(gpp[~s_ok]))
    short = case.status == 0x80
    assert np.array_equal(fpp[short], gpp[short])


def test_packed_pl_input_equals_likelihood_input():
    """PL -> likelihood on the device is the same table the host would use (libm pow), including
    huge PLs (exactly 0), missing samples and unsequenced members (flat {1,1,1})."""
    ped = fs.Pedigree([1, 2, 3, 4, 5, 6], [0, 0, 2, 2, 0, 5], [0, 0, 1, 1, 0, 3], [1, 2, 1, 2, 2, 2],
                      ["a", "NA", "c", "d", "NA", "f"])
    seq = np.array([5, 0, 3, 2], np.int32)  # VCF column order
    rng = np.random.RandomState(11)
    S = 5000
    pl = rng.randint(0, 400, size=(S, 4, 3)).astype(np.uint16)
    pl[np.arange(S), :, rng.randint(0, 3, S)] = 0
    pl[::7, 1] = [0, 3300, 65534]          # beyond the table: exactly 0
    pl[::11, 2] = fs.PL_MISSING            # missing sample
    flags = rng.randint(0, 4, S).astype(np.uint8)
    import math
    table = np.array([math.pow(10.0, -k / 10.0) for k in range(4096)] + [0.0])
    lk = np.ones((S, 6, 3))
    for k, m in enumerate(seq):
        v = table[np.minimum(pl[:, k].astype(np.int64), 4096)]
        miss = np.all(pl[:, k] == fs.PL_MISSING, axis=1)
        v[miss] = 1.0
        lk[:, m] = v
    ctx = fs.Context(fs.make_model(ped), chunk_sites=1777)
    a = ctx.bn_call_batch(seq, lk=lk, flags=flags)
    b = ctx.bn_call_batch(seq, pl16=pl, flags=flags)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)
    assert len(np.unique(a[3])) >= 2  # the mix exercises more than one status
    with pytest.raises(ValueError):
        ctx.bn_call_batch(seq, lk=lk, pl16=pl)
    with pytest.raises(fs.FamseqError):
        ctx.bn_call_batch([0, 0], lk=lk)  # duplicate member
    ctx.close()


@pytest.mark.parametrize("label,opt", CALL_ENGINES, ids=[e[0] for e in CALL_ENGINES])
@pytest.mark.parametrize("cols", [[9, 0, 4], [2], [7, 6, 5, 4, 3, 2, 1, 0], [1, 3, 5, 7, 9, 0, 2]])
def test_caller_names_other_columns_than_the_sequenced_set(label, opt, cols):
    """The generated call-path kernels know the pedigree's number of sequenced members at compile time and walk their output rows
    with that width as a constant; a caller that names another set of columns (fewer samples in this VCF than the model was told are
    sequenced; odd and even counts, one column) takes the walk that reads the width from the arguments.  Whole chunks and a ragged
    one, packed PLs and fp64 rows, text records too; against the plain path's posteriors."""
    ped = fs.synthetic_pedigree("ped10")
    s = 2 * 1024 + 77 if opt.get("enum_impl") != 0 else 300
    mo, fa = ped.relations()
    pl, known, _ = fs.synth.gen_sites(mo, fa, s, seed=fs.synth.SEED_BASE + 7)
    seq = np.array(cols, np.int32)
    k = len(seq)
    lk = np.ones((s, ped.n, 3))
    lk[:, seq] = fs.synth.pl_to_lk(pl[:, seq])
    flags = known.astype(np.uint8)
    ctx = fs.Context(fs.make_model(ped), **opt)
    post, single, st = ctx.bn_batch(lk, flags)
    a = ctx.bn_call_batch(seq, lk=lk, flags=flags)
    b = ctx.bn_call_batch(seq, pl16=np.ascontiguousarray(pl[:, seq]).astype(np.uint16), flags=flags)
    text, st3 = ctx.bn_call_text_batch(seq, pl16=np.ascontiguousarray(pl[:, seq]).astype(np.uint16), flags=flags)
    ctx.close()
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)
    gpp, fpp, fgt, st2 = a
    assert np.array_equal(st, st2) and np.array_equal(st, st3) and not (st & 3).any()
    np.testing.assert_allclose(gpp, host_phred(single[:, seq]), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(fpp, host_phred(post[:, seq]), rtol=1e-12, atol=1e-12)
    assert np.array_equal(fgt, fs.call_genotypes(post[:, seq]).reshape(-1, k))
    for i in (0, 1, 255, 256, 1023, 1024, s - 1):
        if i >= s:
            continue
        for j in range(k):
            rec = text[i, j]
            want = "%g,%g,%g:%g,%g,%g:%s\t" % (tuple(gpp[i, j]) + tuple(fpp[i, j]) + ({0: "0/0", 1: "0/1"}.get(int(fgt[i, j]), "1/1"),))
            assert bytes(rec[:rec[-1]]) == want.encode(), (i, j)


@pytest.mark.parametrize("label,opt", CALL_ENGINES, ids=[e[0] for e in CALL_ENGINES])
def test_any_output_may_be_null(label, opt):
    """famseq_bn_call_batch with some of gpp / fpp / fgt / status NULL (straight through ctypes): what is asked for is what the
    full call returns."""
    import ctypes as C

    ped = fs.synthetic_pedigree("ped10")
    s = 3000 if opt.get("enum_impl") != 0 else 300
    mo, fa = ped.relations()
    pl, known, _ = fs.synth.gen_sites(mo, fa, s, seed=fs.synth.SEED_BASE + 9)
    pl16, flags = np.ascontiguousarray(pl.astype(np.uint16)), known.astype(np.uint8)
    seq = np.arange(ped.n, dtype=np.int32)
    ctx = fs.Context(fs.make_model(ped), **opt)
    full = ctx.bn_call_batch(seq, pl16=pl16, flags=flags)
    L, p = fs.lib(), lambda a, t: a.ctypes.data_as(C.POINTER(t))
    for want in ((0, 0, 1, 0), (1, 0, 0, 1), (0, 1, 0, 0), (0, 1, 1, 1)):
        gpp, fpp = np.full((s, ped.n, 3), -7.0), np.full((s, ped.n, 3), -7.0)
        fgt, st = np.full((s, ped.n), 9, np.int8), np.full(s, 77, np.uint8)
        rc = L.famseq_bn_call_batch(ctx._h, s, None, p(pl16, C.c_uint16), p(flags, C.c_uint8), p(seq, C.c_int32), ped.n,
                                    p(gpp, C.c_double) if want[0] else None, p(fpp, C.c_double) if want[1] else None,
                                    p(fgt, C.c_int8) if want[2] else None, p(st, C.c_uint8) if want[3] else None)
        assert rc == 0
        for got, ref, asked, untouched in ((gpp, full[0], want[0], -7.0), (fpp, full[1], want[1], -7.0), (fgt, full[2], want[2], 9), (st, full[3], want[3], 77)):
            assert np.array_equal(got, ref) if asked else np.all(got == untouched), (label, want)
    ctx.close()
