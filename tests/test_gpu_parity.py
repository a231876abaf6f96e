"""Parity of the HIP path (through the C ABI) with the oracle and the reference fixtures.

Bars (BASELINE.json north_star: posteriors to 1e-6 relative):
  * status byte and single posterior: bit-exact (same arithmetic order, fp64, no FMA);
  * BN posterior: rtol 1e-9 here (the enumeration only re-orders an fp64 sum of
    non-negative terms, so agreement is ~1e-15; 1e-9 leaves margin, 1e-6 is the bar)."""
import numpy as np
import pytest

import famseq_amd as fs
import oracle
from _cases import load_cases

pytestmark = pytest.mark.gpu

RTOL = 1e-9
CASES = load_cases()
BY = {c.name: c for c in CASES}


def check(c, post, single, st, ref=None):
    rpost, rsingle, rst = ref if ref is not None else (c.post, c.single, c.status)
    assert np.array_equal(st, rst)
    ok, s_ok = (rst & 3) == 0, (rst & 3) != 1
    assert np.array_equal(single[s_ok].view(np.uint64), rsingle[s_ok].view(np.uint64))
    np.testing.assert_allclose(post[ok], rpost[ok], rtol=RTOL, atol=0)
    assert np.all(np.isnan(post[~ok])) and np.all(np.isnan(single[~s_ok]))
    short = (rst & 0x80) != 0
    assert np.array_equal(post[short].view(np.uint64), rpost[short].view(np.uint64))


@pytest.mark.parametrize("impl", [0, 1], ids=["team", "lane"])
@pytest.mark.parametrize("c", CASES, ids=repr)
def test_fixture_parity(c, impl):
    """Every compiled-reference fixture: 72 VCF site results, 600 LK rows, synthetic
    autosome/chrX/custom-constant batches, failure and shortcut-boundary probes — through both
    enumeration kernels (team-per-site compiled in, lane-per-site generated per pedigree)."""
    ctx = fs.Context(fs.make_model(c.pedigree(), **c.consts), enum_impl=impl)
    check(c, *ctx.bn_batch(c.lk, c.flags))
    ctx.close()


def elim_ctx(c):
    ctx = fs.Context(fs.make_model(c.pedigree(), **c.consts))
    if not ctx.plan()["elim_supported"]:
        ctx.close()
        pytest.skip("pedigree has loops: enumeration engine only")
    ctx.set_option("engine", fs.ENGINE_ELIM)
    return ctx


@pytest.mark.parametrize("c", CASES, ids=repr)
def test_fixture_parity_elimination_engine(c):
    """The exact sum-product engine (kernel generated per pedigree) against the same
    compiled-reference fixtures, same bars."""
    ctx = elim_ctx(c)
    check(c, *ctx.bn_batch(c.lk, c.flags))
    ctx.close()


def test_elimination_equals_enumeration_on_a_loop_free_20_member_pedigree():
    """N = 20 is out of reach for the oracle (3^20 configs/site) but not for the GPU: both
    engines must agree with each other there, and with the oracle on N = 13."""
    ids, mids, fids, gen = [1, 2], [0, 0], [0, 0], [1, 2]
    while len(ids) < 20:  # a line of descent: each child marries a founder
        child, spouse = len(ids) + 1, len(ids) + 2
        male_child = (len(ids) // 2) % 2 == 0
        ids += [child, spouse]
        prev_child, prev_spouse = ids[-4], ids[-3]
        mo = prev_child if gen[-2] == 2 else prev_spouse
        fa = prev_spouse if gen[-2] == 2 else prev_child
        mids += [mo, 0]
        fids += [fa, 0]
        gen += [1, 2] if male_child else [2, 1]
    ped = fs.Pedigree(ids, mids, fids, gen, ["s%d" % i for i in ids])
    mo, fa = ped.relations()
    for n in (13, 20):
        sub = fs.Pedigree(ids[:n], mids[:n], fids[:n], gen[:n], ped.names[:n])
        mo, fa = sub.relations()
        lk, flags = fs.synth.gen_batch(mo, fa, 4 if n == 13 else 2, 51)
        flags[0] |= 2
        en = fs.Context(fs.make_model(sub), enum_impl=0)
        el = fs.Context(fs.make_model(sub), engine=fs.ENGINE_ELIM)
        a, b = en.bn_batch(lk, flags), el.bn_batch(lk, flags)
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1])
        np.testing.assert_allclose(a[0], b[0], rtol=RTOL)
        if n == 13:
            ln = fs.Context(fs.make_model(sub), enum_impl=1)
            c3 = ln.bn_batch(lk, flags)
            assert np.array_equal(a[2], c3[2]) and np.array_equal(a[1], c3[1])
            np.testing.assert_allclose(a[0], c3[0], rtol=RTOL)
            ln.close()
        if n == 13:
            ref = oracle.OracleModel(sub.ids, sub.mids, sub.fids, sub.genders).bn_batch(lk, flags, threads=4)
            np.testing.assert_allclose(b[0], ref[0], rtol=RTOL)
        en.close()
        el.close()


TILINGS = [dict(fixed_digits=0), dict(fixed_digits=1), dict(fixed_digits=2, low_members=1), dict(low_members=2),
           dict(fixed_digits=3, low_members=2), dict(fixed_digits=6), dict(block_threads=512), dict(block_threads=64)]


@pytest.mark.parametrize("name", ["bn_synth:ped10", "bn_synth:ped10_x", "bn_vcf:fam01", "bn_synth:ped5", "bn_synth:chain7", "bn_synth:quad_mu0"])
@pytest.mark.parametrize("opt", TILINGS, ids=lambda o: ",".join("%s%d" % (k[0], v) for k, v in o.items()))
def test_every_tiling_gives_the_same_answer(name, opt):
    """The result must not depend on how the 3^N space is cut into lanes/steps/low loop
    (exercises multi-team workgroups, 1..3 nested iter levels, 729-lane teams)."""
    c = BY[name]
    try:
        ctx = fs.Context(fs.make_model(c.pedigree(), **c.consts), enum_impl=0, **opt)
    except fs.FamseqError as e:
        pytest.skip("tiling not applicable: %s" % e)
    check(c, *ctx.bn_batch(c.lk, c.flags))
    ctx.close()


def test_null_optional_outputs_and_empty_batch():
    c = BY["bn_synth:ped5"]
    ctx = fs.Context(fs.make_model(c.pedigree()))
    post, single, st = ctx.bn_batch(c.lk, None, want_single=False, want_status=False)
    assert single is None and st is None
    ref = ctx.bn_batch(c.lk, np.zeros(len(c.lk), np.uint8))
    assert np.array_equal(post, ref[0])
    p0, _, s0 = ctx.bn_batch(np.zeros((0, 5, 3)), np.zeros(0, np.uint8))
    assert p0.shape == (0, 5, 3) and s0.shape == (0,)
    ctx.close()


@pytest.mark.parametrize("n_sites", [1, 27, 28, 29, 57, 1000])
def test_ragged_batch_sizes(n_sites):
    """Batch sizes around the 28-sites-per-workgroup tiling of ped5, chunked staging included."""
    c = BY["bn_synth:ped5"]
    reps = -(-n_sites // len(c.lk))
    lk = np.tile(c.lk, (reps, 1, 1))[:n_sites]
    flags = np.tile(c.flags, reps)[:n_sites]
    ref = tuple(np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:n_sites] for a in (c.post, c.single, c.status))
    ctx = fs.Context(fs.make_model(c.pedigree()), chunk_sites=400)
    check(c, *ctx.bn_batch(lk, flags), ref=ref)
    ctx.close()


def test_family_mirror_reads_like_the_reference():
    """set_LK / calPostProbBN / get_postProb / get_postRlt on fam04 (PED order != VCF order)."""
    c = BY["bn_vcf:fam04"]
    ped = fs.read_ped(fs.os.path.join(fs.HERE, "..", "tests", "golden", "testdata", "fam04.ped"))
    fam = fs.Family(ped)
    names = ["ind%02d" % k for k in range(1, 16)]
    fam.set_mapV2P([ped.names.index(n) if n in ped.names else -1 for n in names])
    assert fam.get_numInd() == 6 and fam.get_realNumInd() == 2
    assert fam.set_LK(c.lk)
    ok = fam.calPostProbBN((c.flags & 1) != 0, (c.flags >> 1) & 1)
    assert ok.all()
    seq = [ped.names.index("ind08"), ped.names.index("ind09")]  # VCF column order
    np.testing.assert_allclose(fam.get_postProb(), c.post[:, seq, :], rtol=RTOL)
    assert np.array_equal(fam.get_postProbSingle(), c.single[:, seq, :])
    want = np.array([[oracle.argmax3(r) for r in site] for site in c.post[:, seq, :]])
    assert np.array_equal(fam.get_postRlt(), want)
    assert not fam.set_LK(np.ones((3, 5, 3)))


def test_device_pointer_entry_matches_host_entry():
    torch = pytest.importorskip("torch")
    c = BY["bn_synth:ped10"]
    ctx = fs.Context(fs.make_model(c.pedigree()))
    dev = torch.device("cuda:0")
    lk = torch.from_numpy(c.lk).to(dev)
    fl = torch.from_numpy(c.flags).to(dev)
    post, single = torch.empty_like(lk), torch.empty_like(lk)
    st = torch.empty(len(c.lk), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    ctx.bn_batch_device(len(c.lk), lk.data_ptr(), fl.data_ptr(), post.data_ptr(), single.data_ptr(), st.data_ptr(),
                        stream.cuda_stream)
    stream.synchronize()
    host = ctx.bn_batch(c.lk, c.flags)
    assert np.array_equal(post.cpu().numpy(), host[0]) and np.array_equal(single.cpu().numpy(), host[1])
    assert np.array_equal(st.cpu().numpy(), host[2])
    ctx.close()


@pytest.mark.parametrize("engine", [dict(enum_impl=1), dict(engine=1)], ids=["lane", "elim"])
@pytest.mark.parametrize("name,cfg", [("ped5", 1), ("ped10", 2)])
def test_device_arrays_that_are_only_8_byte_aligned(name, cfg, engine):
    """The generated kernels move 16 B per lane when lk/post/single are 16-byte aligned and fall
    back to 8 B per lane otherwise (decided per launch).  Both paths, whole chunks plus a ragged
    tail, must give the same bits."""
    torch = pytest.importorskip("torch")
    ped = fs.synthetic_pedigree(name)
    mo, fa = ped.relations()
    n_sites = 3 * 256 + 77
    lk_h, fl_h = fs.synth.gen_batch(mo, fa, n_sites, cfg)
    ctx = fs.Context(fs.make_model(ped), **engine)
    dev = torch.device("cuda:0")
    w = lk_h.size
    res = []
    for off in (0, 1):  # element offset into an over-allocated buffer: 0 -> 16-B aligned, 1 -> 8-B aligned
        bufs = [torch.zeros(w + 2, dtype=torch.float64, device=dev) for _ in range(3)]
        lk, post, single = (b[off:off + w] for b in bufs)
        assert lk.data_ptr() % 16 == 8 * off
        lk.copy_(torch.from_numpy(lk_h.reshape(-1)))
        fl = torch.from_numpy(fl_h).to(dev)
        st = torch.empty(n_sites, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream()
        ctx.bn_batch_device(n_sites, lk.data_ptr(), fl.data_ptr(), post.data_ptr(), single.data_ptr(), st.data_ptr(),
                            stream.cuda_stream)
        stream.synchronize()
        res.append((post.cpu().numpy().copy(), single.cpu().numpy().copy(), st.cpu().numpy().copy()))
        for b in bufs:  # nothing written outside the window
            assert float(b[:off].abs().sum()) == 0 and float(b[off + w:].abs().sum()) == 0 or b is bufs[0]
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    host = ctx.bn_batch(lk_h, fl_h)
    assert np.array_equal(res[0][0].reshape(host[0].shape), host[0])
    assert np.array_equal(res[0][2], host[2])
    ctx.close()


@pytest.mark.parametrize("engine", [dict(enum_impl=0), dict(enum_impl=1), dict(engine=1)], ids=["team", "lane", "elim"])
@pytest.mark.parametrize("name,cfg,n_sites", [("ped5", 1, 200_000), ("ped10", 2, 20_000)])
def test_full_size_properties(name, cfg, n_sites, engine):
    """Size-independent properties on large seeded batches (BASELINE configs' shape):
    rows are distributions, the run is bit-reproducible, the answer does not depend on
    batch position or on how the batch is sharded, and a sampled subset matches the oracle."""
    ped = fs.synthetic_pedigree(name)
    mo, fa = ped.relations()
    lk, flags = fs.synth.gen_batch(mo, fa, n_sites, cfg)
    ctx = fs.Context(fs.make_model(ped), **engine)
    post, single, st = ctx.bn_batch(lk, flags)
    assert np.all(st == 0)  # generator guarantees no shortcut, no failure
    assert np.all(post >= 0) and np.allclose(post.sum(axis=2), 1.0, rtol=0, atol=1e-12)
    again = ctx.bn_batch(lk, flags)
    assert np.array_equal(again[0], post) and np.array_equal(again[1], single)
    perm = np.random.RandomState(3).permutation(n_sites)
    p2 = ctx.bn_batch(lk[perm], flags[perm])[0]
    assert np.array_equal(p2, post[perm])
    cut = n_sites // 3 + 1
    halves = [ctx.bn_batch(lk[a:b], flags[a:b])[0] for a, b in ((0, cut), (cut, n_sites))]
    assert np.array_equal(np.concatenate(halves), post)
    idx = np.random.RandomState(5).choice(n_sites, 64, replace=False)
    o = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders)
    ref = o.bn_batch(lk[idx], flags[idx], threads=4)
    np.testing.assert_allclose(post[idx], ref[0], rtol=RTOL)
    assert np.array_equal(single[idx], ref[1])
    ctx.close()


def test_deep_enumeration_ped15_and_max_members():
    """3^15 configurations per site (BASELINE config 5 shape) and a 13-member chain with a
    single childless member (two nested iter levels); one or two sites against the oracle."""
    ped = fs.synthetic_pedigree("ped15")
    mo, fa = ped.relations()
    lk, flags = fs.synth.gen_batch(mo, fa, 2, 5)
    ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders).bn_batch(lk, flags, threads=2)
    for impl in (0, 1):
        ctx = fs.Context(fs.make_model(ped), enum_impl=impl)
        post, single, st = ctx.bn_batch(lk, flags)
        assert np.array_equal(st, ref[2])
        np.testing.assert_allclose(post, ref[0], rtol=RTOL)
        ctx.close()
    ids = list(range(1, 14))
    mids = [0, 0, 2, 0, 4, 0, 6, 0, 8, 0, 10, 0, 12]
    fids = [0, 0, 1, 0, 3, 0, 5, 0, 7, 0, 9, 0, 11]
    gen = [1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 2]
    chain = fs.Pedigree(ids, mids, fids, gen, ["s%d" % i for i in ids])
    mo, fa = chain.relations()
    lk, flags = fs.synth.gen_batch(mo, fa, 3, 41)
    flags[1] |= 2
    ctx = fs.Context(fs.make_model(chain), enum_impl=0)
    assert ctx.plan()["jlevels"] == 2
    post, single, st = ctx.bn_batch(lk, flags)
    ref = oracle.OracleModel(ids, mids, fids, gen).bn_batch(lk, flags, threads=3)
    assert np.array_equal(st, ref[2])
    np.testing.assert_allclose(post, ref[0], rtol=RTOL)
    ctx.close()


def test_degenerate_pedigrees():
    """One member; two unrelated members; a trio whose only sequenced member is the child."""
    for ids, mids, fids, gen, names in (([1], [0], [0], [1], ["a"]),
                                        ([1, 2], [0, 0], [0, 0], [2, 1], ["a", "b"]),
                                        ([1, 2, 3], [0, 0, 2], [0, 0, 1], [1, 2, 1], ["NA", "NA", "c"])):
        ped = fs.Pedigree(ids, mids, fids, gen, names)
        mo, fa = ped.relations()
        lk, flags = fs.synth.gen_batch(mo, fa, 40, 61)
        lk[:, ped.sequenced == 0, :] = 1.0
        flags[::3] |= 2
        ref = oracle.OracleModel(ids, mids, fids, gen, ped.sequenced).bn_batch(lk, flags)
        for opt in (dict(enum_impl=0), dict(enum_impl=1), dict(engine=fs.ENGINE_ELIM)):
            ctx = fs.Context(fs.make_model(ped), **opt)
            post, single, st = ctx.bn_batch(lk, flags)
            ctx.close()
            assert np.array_equal(st, ref[2]) and np.array_equal(single, ref[1])
            np.testing.assert_allclose(post, ref[0], rtol=RTOL, atol=0)


def test_non_finite_and_negative_likelihoods_follow_the_reference():
    """Garbage in.  Branch decisions (status byte) and the single posterior are the same
    statements as the CPU code (`s <= 0` is false for NaN, `big < lc` is false for NaN, ...), so
    they must agree bit for bit even for NaN / Inf / negative likelihoods.  The BN posterior is
    compared only where the inputs are finite: with Inf or NaN among the factors, which marginals
    turn NaN depends on the order of the products (inf * 0), which the device re-orders by design."""
    c = BY["bn_synth:quad"]
    rng = np.random.RandomState(3)
    lk = np.tile(c.lk[:1], (48, 1, 1)).copy()
    lk[:, 3] = 10.0 ** (-rng.randint(0, 120, size=(48, 3)) / 10.0)  # soft child: full enumeration unless broken below
    bad = [np.nan, np.inf, -np.inf, -1.0, 0.0, 1e300]
    for s in range(48):
        i, g = rng.randint(0, 4), rng.randint(0, 3)
        lk[s, i, g] = bad[s % 6]
    finite_in = np.all(np.isfinite(lk), axis=(1, 2)) & np.all(lk >= 0, axis=(1, 2))
    flags = (np.arange(48) % 4).astype(np.uint8)
    ref = oracle.OracleModel(c.ids, c.mids, c.fids, c.genders, c.sequenced).bn_batch(lk, flags)
    for opt in (dict(enum_impl=0), dict(enum_impl=1), dict(engine=fs.ENGINE_ELIM)):
        ctx = fs.Context(fs.make_model(c.pedigree()), **opt)
        post, single, st = ctx.bn_batch(lk, flags)
        ctx.close()
        assert np.array_equal(st, ref[2]), opt
        assert np.array_equal(single, ref[1], equal_nan=True), opt
        ok = finite_in & ((st & 3) == 0)
        assert ok.sum() >= 10
        np.testing.assert_allclose(post[ok], ref[0][ok], rtol=RTOL, atol=0, err_msg=str(opt))
        assert np.all(np.isnan(post[(st & 3) != 0]))


@pytest.mark.parametrize("engine", [dict(enum_impl=0), dict(enum_impl=1), dict(engine=1)], ids=["team", "lane", "elim"])
def test_known_and_chrx_bits_mixed_inside_every_wave(engine, monkeypatch):
    """flags cycle through all four (Known, chrX) combinations from site to site, so every wave of
    every kernel holds all of them (the sum-product kernel runs its body once per chrX value
    present in a wave, with a wave-uniform table pointer)."""
    ped = fs.synthetic_pedigree("ped10")
    mo, fa = ped.relations()
    n_sites = 700
    lk, _ = fs.synth.gen_batch(mo, fa, n_sites, 2)
    flags = (np.arange(n_sites) % 4).astype(np.uint8)
    flags[100:164] = 2  # one wave that is uniform in chrX = 1
    ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders).bn_batch(lk, flags)
    if "engine" in engine:
        monkeypatch.setenv("FAMSEQ_VARIANT_MIN", "1")  # the variants with the per-chrX passes (the picker starts from the fence-free one)
    ctx = fs.Context(fs.make_model(ped), **engine)
    post, single, st = ctx.bn_batch(lk, flags)
    if "engine" in engine:
        assert ctx.plan()["elim_variant"] >= 1
    ctx.close()
    assert np.array_equal(st, ref[2]) and np.array_equal(single, ref[1])
    np.testing.assert_allclose(post, ref[0], rtol=RTOL, atol=0)


def test_sum_product_on_a_pedigree_with_a_loop():
    """First-cousin marriage: the sum-product engine conditions on one member; same answer as both
    enumeration kernels and the oracle."""
    from test_elim import cousins_marry
    from test_gpu_random_pedigrees import random_likelihoods

    ped = cousins_marry()
    ped.relations()
    lk, flags = random_likelihoods(np.random.RandomState(11), ped, 300)
    ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders, ped.sequenced).bn_batch(lk, flags, threads=4)
    for opt in (dict(engine=fs.ENGINE_ELIM), dict(enum_impl=1), dict(enum_impl=0)):
        ctx = fs.Context(fs.make_model(ped), **opt)
        if "engine" in opt:
            assert ctx.plan()["elim_conditioned_members"] == 1
        post, single, st = ctx.bn_batch(lk, flags)
        ctx.close()
        assert np.array_equal(st, ref[2]), opt
        ok = (st & 3) == 0
        np.testing.assert_allclose(post[ok], ref[0][ok], rtol=RTOL, atol=0, err_msg=str(opt))


def test_sharded_entry_point_over_two_contexts():
    """famseq_bn_batch_sharded: contiguous site ranges over several contexts, one host thread each
    (two contexts on the one GPU of this box), same bits as a single call; errors come back."""
    c = BY["bn_synth:ped10"]
    model = fs.make_model(c.pedigree())
    lk = np.tile(c.lk, (40, 1, 1))[:1501]
    flags = np.tile(c.flags, 40)[:1501]
    # one kernel on both sides (auto mode picks the lanes per site from the batch size, and a different
    # summation order shows in the last bits)
    one = fs.Context(model, enum_impl=0)
    ref = one.bn_batch(lk, flags)
    one.close()
    ctxs = [fs.Context(model, enum_impl=0), fs.Context(model, enum_impl=0), fs.Context(model, enum_impl=0)]
    got = fs.bn_batch_sharded(ctxs, lk, flags)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b, equal_nan=True)
    empty = fs.bn_batch_sharded(ctxs, lk[:0], flags[:0])
    assert empty[0].shape[0] == 0
    for x in ctxs:
        x.close()
    with pytest.raises(fs.FamseqError, match="no CPU path"):
        fs.bn_batch_sharded([fs.Context(model, device=-1)], lk[:4], flags[:4])


@pytest.mark.parametrize("name,n_sites", [("trio", 600_000), ("quad", 400_000), ("ped5", 300_000), ("ped10", 20_000)])
def test_single_posterior_quotients_are_the_divisions_bits(name, n_sites):
    """The generated kernels form the single posterior's three quotients per member with ONE refined reciprocal where every
    intermediate is a normal number (elim_codegen.cpp kDiv3Text: the compiler's own division sequence with its scaling steps
    left out where they are the identity) and with plain divisions elsewhere; the compiled-in team kernel divides.  Likelihoods
    over the whole exponent range — down to the sub-normals, exact zeros, values above 1 (an LK file's scale), a member with all
    three tiny — every flag combination: the same bits, site for site, from the enumeration and the sum-product kernel."""
    ped = fs.synthetic_pedigree(name)
    rng = np.random.RandomState(42)
    lk = 10.0 ** (-rng.uniform(0, 300, size=(n_sites, ped.n, 3)) * (rng.random_sample((n_sites, ped.n, 1)) < 0.5))
    lk[rng.random_sample(lk.shape) < 0.01] = 0.0
    lk[rng.random_sample(lk.shape) < 0.01] *= 1e6
    tiny = rng.random_sample((n_sites, ped.n)) < 0.01
    lk[tiny] = lk[tiny] * 1e-305
    huge = rng.random_sample((n_sites, ped.n)) < 0.002
    lk[huge] = lk[huge] * 1e302
    flags = rng.randint(0, 4, n_sites).astype(np.uint8)
    model = fs.make_model(ped)
    want = None
    for label, opt in (("team", dict(enum_impl=0)), ("lane", dict(enum_impl=1)), ("elim", dict(engine=fs.ENGINE_ELIM))):
        if label == "team" and name == "ped10":
            continue
        ctx = fs.Context(model, **opt)
        _, single, st = ctx.bn_batch(lk[:n_sites if label != "team" else min(n_sites, 100_000)], flags[:n_sites if label != "team" else min(n_sites, 100_000)])
        ctx.close()
        if want is None:
            want = (single, st)
            continue
        k = min(len(st), len(want[1]))
        assert np.array_equal(st[:k] & 1, want[1][:k] & 1), label
        ok = (st[:k] & 3) != 1
        assert np.array_equal(single[:k][ok].view(np.uint64), want[0][:k][ok].view(np.uint64)), label
    # and against numpy's IEEE divisions of the same products (family.cpp:1426-1445), on the sites whose single posterior exists
    single, st = want
    k = len(st)
    m = model
    autos = (flags[:k] & 2) == 0
    prior = np.where(((flags[:k] & 1) != 0)[:, None], np.array(list(m.genoProbK))[None, :], np.array(list(m.genoProbN))[None, :])
    p = lk[:k] * prior[:, None, :]
    s = (p[:, :, 0] + p[:, :, 1]) + p[:, :, 2]
    with np.errstate(all="ignore"):
        ref = p / s[:, :, None]
    sel = autos & ((st & 3) != 1)
    assert sel.sum() > 1000
    assert np.array_equal(single[sel].view(np.uint64), ref[sel].view(np.uint64))
