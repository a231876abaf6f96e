"""Pedigrees of more than 20 members (the domain the reference recommends -method 2 for: family.cpp:1126-1403,
FamSeq_Manual.pdf p.1) — what can be checked without a GPU: the numpy sum-product oracle against the compiled
reference's goldens, the size-independent model at the C ABI, and the generated sum-product kernel's arithmetic
(its source compiled for the host, as tests/test_generated_host.py does) against the same goldens."""
import numpy as np
import pytest

import famseq_amd as fs
import oracle.sum_product as sp
from _wide import SIZES, WideCase, check


@pytest.mark.parametrize("n", SIZES)
def test_numpy_sum_product_oracle_is_pinned_to_the_reference(n):
    case = WideCase(n)
    mo, fa = case.ped.relations()
    assert sp.is_forest(mo, fa)
    post, single, status = sp.pedigree_posterior(case.ped, case.lk, case.flags)
    check(case, post, single, status, rtol=1e-12)
    assert {0, 1, 0x80} <= set(int(s) for s in case.status) and set(int(f) for f in case.flags & 3) == {0, 1, 2, 3}


def test_oracle_agrees_with_the_enumeration_oracle_where_both_apply():
    """... and on a pedigree small enough for bn_oracle.c's 3^N enumeration the two oracles agree."""
    import oracle

    rng = np.random.RandomState(5)
    ped = fs.synth.grow_pedigree(rng, 9, allow_loops=False)
    lk, flags = fs.synth.random_likelihoods(rng, ped, 64)
    ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders, ped.sequenced).bn_batch(lk, flags, threads=2)
    post, single, status = sp.pedigree_posterior(ped, lk, flags)
    assert np.array_equal(status, ref[2])
    ok = (status & 3) == 0
    np.testing.assert_allclose(post[ok], ref[0][ok], rtol=1e-11, atol=0)


def test_size_independent_model_at_the_abi():
    ped = WideCase(24).ped
    m = fs.make_model(ped)
    assert isinstance(m, fs.CPedigree) and m.n_members == 24
    ctx = fs.Context(m, device=-1)  # plan-only: generates and compiles the sum-product kernel
    plan = ctx.plan()
    assert plan["N"] == 24 and plan["engine"] == fs.ENGINE_ELIM and plan["enum_supported"] == 0 and plan["elim_code_object"]
    with pytest.raises(fs.FamseqError, match="20 members"):
        ctx.set_option("engine", fs.ENGINE_ENUM)
    with pytest.raises(fs.FamseqError, match="enumeration"):
        ctx.set_option("enum_impl", 1)
    with pytest.raises(fs.FamseqError):  # no device, no CPU path
        ctx.bn_batch(np.ones((1, 24, 3)))
    ctx.close()
    # a small pedigree through the same struct is exactly famseq_create
    small = fs.synthetic_pedigree("ped5")
    a, b = fs.make_model(small), fs.make_model(small, size_independent=True)
    ca, cb = fs.Context(a, device=-1), fs.Context(b, device=-1)
    pa, pb = ca.plan(), cb.plan()
    assert pa == pb and isinstance(b, fs.CPedigree)
    ca.close(), cb.close()
    # the fixed struct stays at 20 members
    import ctypes as C

    big = fs.CModel()
    big.n_members = 21
    err = C.create_string_buffer(256)
    assert not fs.lib().famseq_create(C.byref(big), -1, err, len(err)) and b"famseq_pedigree" in err.value


def test_wide_pedigree_with_loops_is_refused_with_a_reason():
    rng = np.random.RandomState(3)
    for _ in range(50):
        ped = fs.synth.grow_pedigree(rng, 26, allow_loops=True)
        mo, fa = ped.relations()
        if not sp.is_forest(mo, fa):
            break
    probe_ok = True
    try:
        fs.Context(fs.make_model(ped), device=-1).close()
    except fs.FamseqError as e:
        probe_ok = False
        assert "sum-product" in str(e)
    # (a loop that three conditioned members cut is served; more are refused: either way no crash, and a reason)
    assert probe_ok in (True, False)


@pytest.mark.parametrize("n", SIZES)
def test_generated_kernel_arithmetic_matches_the_reference_goldens(n, tmp_path, monkeypatch):
    from test_generated_host import build_host_kernel, run_host

    case = WideCase(n)
    model = fs.make_model(case.ped)
    fn = build_host_kernel(model, "elim", tmp_path, monkeypatch)
    post, single, st = run_host(fn, model, case.lk, case.flags)
    check(case, post, single, st, rtol=1e-10)


def test_a_pedigree_wider_than_lds_rows_could_stage(tmp_path, monkeypatch):
    """128 members: 3N doubles per lane would be 197 KB of LDS per wave — the generated kernel is the form without LDS
    staging (variant 8), here compiled for the host and checked against the numpy oracle (itself pinned to the reference's
    -method 2 on the 24 / 32 / 48-member goldens)."""
    from famseq_amd.prebuild_sets import wide_pedigree
    from test_generated_host import build_host_kernel, run_host

    ped = wide_pedigree(128)
    rng = np.random.RandomState(128)
    pl = rng.randint(0, 120, size=(40, ped.n, 3)).astype(float)
    pl[np.arange(40)[:, None], np.arange(ped.n)[None, :], rng.randint(0, 3, size=(40, ped.n))] = 0
    lk = 10.0 ** (-pl / 10.0)
    lk[:, ped.sequenced == 0, :] = 1.0
    flags = rng.randint(0, 4, 40).astype(np.uint8)
    want = sp.pedigree_posterior(ped, lk, flags)
    assert np.all(want[2] == 0)
    model = fs.make_model(ped)
    probe = fs.Context(model, device=-1)
    assert probe.plan()["elim_variant"] >= 8
    probe.close()
    fn = build_host_kernel(model, "elim", tmp_path, monkeypatch)
    post, single, st = run_host(fn, model, lk, flags)
    assert np.array_equal(st, want[2]) and np.array_equal(single, want[1])
    np.testing.assert_allclose(post, want[0], rtol=1e-9, atol=1e-300)

