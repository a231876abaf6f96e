"""The CPU oracle (oracle/bn_oracle.c) against the compiled-reference fixtures.

Fixtures were produced by oracle/gen_golden.py from /root/reference's own
family.cpp (oracle/_ref).  Bar: BIT-EXACT posteriors, single posteriors and
status (same arithmetic order, no FMA contraction)."""
import numpy as np
import pytest

import oracle
from _cases import GOLDEN, load_cases

CASES = load_cases()


def model_for(c):
    return oracle.OracleModel(c.ids, c.mids, c.fids, c.genders, c.sequenced, **c.consts)


def same_bits(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


@pytest.mark.parametrize("c", CASES, ids=repr)
def test_oracle_bit_exact(c):
    post, single, st = model_for(c).bn_batch(c.lk, c.flags)
    assert np.array_equal(st, c.status)
    ok = (st & 3) == 0
    assert same_bits(single[(st & 3) != 1], c.single[(c.status & 3) != 1])
    assert same_bits(post[ok], c.post[ok])
    # failed sites are NaN-filled, never stale numbers
    assert np.all(np.isnan(post[~ok]))
    assert np.all(np.isnan(single[(st & 3) == 1]))


def test_fixture_coverage():
    """The pins the survey lists: 72 VCF site results (12 full-BN), 600 LK rows,
    chrX, both failure statuses, shortcut boundary."""
    by = {c.name: c for c in CASES}
    vcf = [c for c in CASES if c.name.startswith("bn_vcf")]
    assert sum(len(c.status) for c in vcf) == 72
    assert sum(int(np.sum((c.status & 0x80) == 0)) for c in vcf) == 12
    assert sum(len(c.status) for c in CASES if c.name.startswith("bn_lk")) == 600
    assert all(np.all(c.status == 0) for c in CASES if c.name.startswith("bn_lk"))
    assert 1 in by["bn_synth:quad"].status and 2 in by["bn_synth:quad_mu0"].status
    assert np.any(by["bn_synth:ped10_x"].flags & 2)
    lrc = by["bn_synth:trio_lrc"].status
    assert 0 < np.count_nonzero(lrc & 0x80) < len(lrc)


@pytest.mark.parametrize("c", [c for c in CASES if c.peel is not None], ids=repr)
def test_enumeration_agrees_with_peeling(c):
    """-method 2 (Elston-Stewart) is an independent analytic cross-check of the
    enumeration on loop-free pedigrees."""
    ok = (c.status & 3) == 0
    np.testing.assert_allclose(c.post[ok], c.peel[ok], rtol=1e-9, atol=1e-300)


def test_tables_bit_exact():
    z = np.load(GOLDEN + "/tables.npz")
    mus = sorted({k.split(".pcp2")[0] for k in z.files if k.endswith(".pcp2")})
    assert len(mus) >= 5
    for m in mus:
        a, b, c = oracle.tables(float(m[2:]))
        assert same_bits(a, z[m + ".pcp2"]) and same_bits(b, z[m + ".xf"]) and same_bits(c, z[m + ".xm"])


def test_threads_match_single_thread():
    c = {c.name: c for c in CASES}["bn_synth:ped5"]
    m = model_for(c)
    a = m.bn_batch(c.lk, c.flags, threads=1)
    b = m.bn_batch(c.lk, c.flags, threads=4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_argmax_first_max_wins():
    assert oracle.argmax3([0.2, 0.5, 0.3]) == 1
    assert oracle.argmax3([0.5, 0.5, 0.0]) == 0  # ties -> lowest genotype (family.cpp:654)
    assert oracle.argmax3([0.0, 0.0, 0.0]) == 0
    assert oracle.argmax3([float("nan")] * 3) == -1


@pytest.mark.skipif(not oracle.have_ref(), reason="compiled reference not present (GPU box)")
def test_oracle_vs_live_reference_random():
    """Fresh random inputs straight through the compiled reference (build container only)."""
    rng = np.random.RandomState(7)
    ped = dict(ids=[1, 2, 3, 4, 5, 6], mids=[0, 0, 2, 2, 0, 5], fids=[0, 0, 1, 1, 0, 3], genders=[1, 2, 1, 2, 2, 2])
    seq = np.array([1, 0, 1, 1, 0, 1], np.uint8)
    lk = 10.0 ** (-rng.randint(0, 400, size=(40, 6, 3)) / 10.0)
    lk[:, seq == 0, :] = 1
    flags = rng.randint(0, 4, 40).astype(np.uint8)
    for kw in ({}, dict(mrate=1e-4, lc=0.999)):
        r = oracle.RefFamily(ped["ids"], ped["mids"], ped["fids"], ped["genders"], seq, **kw).bn_batch(lk, flags)
        o = oracle.OracleModel(ped["ids"], ped["mids"], ped["fids"], ped["genders"], seq, **kw).bn_batch(lk, flags)
        assert np.array_equal(r[2], o[2])
        ok = (r[2] & 3) == 0
        assert same_bits(r[0][ok], o[0][ok]) and same_bits(r[1][ok], o[1][ok])
