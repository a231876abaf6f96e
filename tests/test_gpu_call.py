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
