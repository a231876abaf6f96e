"""Pedigrees of more than 20 members on the GPU, through the C ABI (famseq_create_pedigree + the batch entry points)
and through bin/FamSeq -method 2, against the compiled reference's -method 2 goldens (family::calPostProbPeeling,
family.cpp:1126-1403; tests/golden/wide_peds.npz, tests/golden/ref_cli/wide*) and, on random pedigrees, against the
numpy sum-product oracle pinned to them (oracle/sum_product.py)."""
import os
import subprocess

import numpy as np
import pytest

import famseq_amd as fs
from _wide import SIZES, WideCase, check
from famseq_amd.prebuild_sets import WIDE_RANDOM_SEEDS, wide_random_pedigree

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bin", "FamSeq")
TD = os.path.join(ROOT, "tests", "golden", "testdata")
REF = os.path.join(ROOT, "tests", "golden", "ref_cli")


@pytest.mark.parametrize("n", SIZES)
def test_reference_method2_goldens(n):
    case = WideCase(n)
    ctx = fs.Context(fs.make_model(case.ped), device=0)
    assert ctx.plan()["engine"] == fs.ENGINE_ELIM
    post, single, status = ctx.bn_batch(case.lk, case.flags)
    check(case, post, single, status)
    # ragged sizes around the 64-site chunk, and a repeat (slots reused)
    for s in (1, 63, 65):
        reps = -(-s // len(case.lk))
        lk, fl = np.tile(case.lk, (reps, 1, 1))[:s], np.tile(case.flags, reps)[:s]
        p2, s2, st2 = ctx.bn_batch(lk, fl)
        want = np.tile(case.status, reps)[:s]
        assert np.array_equal(st2, want)
        ok = (want & 3) == 0
        np.testing.assert_allclose(p2[ok], np.tile(case.post, (reps, 1, 1))[:s][ok], rtol=1e-9, atol=0)
    ctx.close()


@pytest.mark.parametrize("seed", WIDE_RANDOM_SEEDS)
def test_random_wide_pedigrees_against_the_pinned_oracle(seed):
    import oracle.sum_product as sp

    rng, ped, mu = wide_random_pedigree(seed)
    lk, flags = fs.synth.random_likelihoods(rng, ped, 300, max_pl=60)  # (see its docstring: 300 would put sites below 1e-308)
    want = sp.pedigree_posterior(ped, lk, flags, mrate=mu)
    ctx = fs.Context(fs.make_model(ped, mrate=mu), device=0)
    post, single, status = ctx.bn_batch(lk, flags)
    ctx.close()
    # These inputs are adversarial by construction (members contradict their parents at random): with fifty of them a site's
    # whole probability mass can sit near 1e-308, where double arithmetic — the reference's, the oracle's, the kernel's, each in
    # its own order of products — keeps only what gradual underflow leaves, or reaches zero and fails the site (status 2).  The
    # same message passing in x87 long double (15-bit exponent) says which sites those are: the comparison is made on the sites
    # where the oracle's double arithmetic reproduces its own long-double result, and those must be most of the batch.
    true = sp.pedigree_posterior(ped, lk, flags, mrate=mu, dtype=np.longdouble)
    sound = (want[2] == true[2]) & np.all(np.isclose(want[0], true[0].astype(np.float64), rtol=1e-10, atol=1e-35, equal_nan=True), axis=(1, 2))
    assert sound.mean() > 0.8
    assert np.array_equal(status[sound], want[2][sound])
    ok, s_ok = sound & ((want[2] & 3) == 0), (want[2] & 3) != 1
    assert np.array_equal(single[s_ok], want[1][s_ok]) and np.array_equal(status[~s_ok], want[2][~s_ok])
    np.testing.assert_allclose(post[ok], want[0][ok], rtol=1e-9, atol=1e-35)
    # elsewhere: a status the reference's family of answers allows, and rows that are NaN or a distribution
    assert np.all(np.isin(status[~sound], (0, 2)))
    loose = ~sound & (status == 0)
    assert np.all(np.isfinite(post[loose])) and np.all(np.abs(post[loose].sum(axis=2) - 1) < 1e-6)
    assert np.all(np.isnan(post[(status & 3) != 0]))


def test_a_pedigree_wider_than_lds_rows_could_stage():
    """128 members through the kernel form without LDS staging, ragged batch, against the numpy oracle."""
    import oracle.sum_product as sp
    from famseq_amd.prebuild_sets import wide_pedigree

    ped = wide_pedigree(128)
    rng = np.random.RandomState(128)
    s = 1000
    pl = rng.randint(0, 120, size=(s, ped.n, 3)).astype(float)
    pl[np.arange(s)[:, None], np.arange(ped.n)[None, :], rng.randint(0, 3, size=(s, ped.n))] = 0
    lk = 10.0 ** (-pl / 10.0)
    lk[:, ped.sequenced == 0, :] = 1.0
    lk[7] = 1.0
    lk[7, ped.sequenced == 1] = [1.0, 1e-17, 1e-20]  # a shortcut site
    lk[9, int(np.nonzero(ped.sequenced)[0][3])] = 0.0  # a site whose single posterior fails
    flags = rng.randint(0, 4, s).astype(np.uint8)
    want = sp.pedigree_posterior(ped, lk, flags)
    ctx = fs.Context(fs.make_model(ped), device=0)
    assert ctx.plan()["elim_variant"] >= 8
    post, single, status = ctx.bn_batch(lk, flags)
    ctx.close()
    assert np.array_equal(status, want[2]) and status[7] == 0x80 and status[9] == 1
    ok, s_ok = (status & 3) == 0, (status & 3) != 1
    assert np.array_equal(single[s_ok], want[1][s_ok])
    np.testing.assert_allclose(post[ok], want[0][ok], rtol=1e-9, atol=1e-300)
    assert np.all(np.isnan(post[~ok])) and np.all(np.isnan(single[~s_ok]))


def test_call_path_of_a_wide_pedigree():
    """famseq_bn_call_batch (packed PLs in, GPP / FPP / FGT out) on the 32-member pedigree: separate unpack /
    posterior / Phred stages (no fused form beyond 20 members), against the drivers' formulas (file.cpp:696-745,
    family.cpp:636-665) applied to the oracle's posteriors."""
    import math

    import oracle.sum_product as sp

    case = WideCase(32)
    ped = case.ped
    seq = np.nonzero(ped.sequenced)[0].astype(np.int32)
    rng = np.random.RandomState(11)
    s = 500
    pl = rng.randint(0, 256, size=(s, len(seq), 3)).astype(np.uint16)
    pl[np.arange(s)[:, None], np.arange(len(seq))[None, :], rng.randint(0, 3, size=(s, len(seq)))] = 0
    pl[5, 2] = fs.PL_MISSING
    flags = rng.randint(0, 4, s).astype(np.uint8)
    table = np.array([math.pow(10.0, -k / 10.0) for k in range(256)])
    lk = np.ones((s, ped.n, 3))
    lk[:, seq, :] = table[pl.astype(np.int64) % 256]
    lk[5, seq[2]] = 1.0
    want = sp.pedigree_posterior(ped, lk, flags)
    ctx = fs.Context(fs.make_model(ped), device=0)
    gpp, fpp, fgt, st = ctx.bn_call_batch(seq, pl16=pl, flags=flags)
    ctx.close()
    assert np.array_equal(st, want[2])
    ok = (st & 3) == 0
    with np.errstate(divide="ignore"):
        f_want = np.abs(-10 * np.log10(want[0][:, seq, :]))
        g_want = np.abs(-10 * np.log10(want[1][:, seq, :]))
    f_want[np.isinf(f_want)] = 99999.0
    g_want[np.isinf(g_want)] = 99999.0
    np.testing.assert_allclose(fpp[ok], f_want[ok], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(gpp[ok], g_want[ok], rtol=1e-12, atol=1e-12)
    best = np.argmax(want[0][:, seq, :], axis=2)
    p_sorted = np.sort(want[0][:, seq, :], axis=2)
    clear = ok[:, None] & (p_sorted[:, :, 2] - p_sorted[:, :, 1] > 1e-9)
    assert np.array_equal(fgt[clear], best[clear])


def run_cli(args, out):
    p = subprocess.run([CLI] + args + ["-output", str(out)], capture_output=True, text=True, timeout=300)
    return p.returncode, p.stdout + p.stderr


def test_cli_method2_on_wide_pedigrees(tmp_path):
    from test_cli_gpu import assert_same_output

    out = tmp_path / "w32.vcf"
    rc, msg = run_cli(["vcf", "-vcfFile", TD + "/wide32.vcf", "-pedFile", TD + "/wide32.ped", "-method", "2", "-v"], out)
    assert rc == 0, msg
    assert assert_same_output(out, REF + "/wide32_method2.vcf") == 60
    out = tmp_path / "w48.txt"
    rc, msg = run_cli(["LK", "-lkFile", TD + "/wide48_lk.txt", "-pedFile", TD + "/wide48.ped", "-method", "2"], out)
    assert rc == 0, msg
    assert assert_same_output(out, REF + "/wide48_lk_method2.txt", tags="LK:GPP") >= 40


def test_cli_method1_on_a_wide_pedigree_says_what_to_do(tmp_path):
    rc, msg = run_cli(["vcf", "-vcfFile", TD + "/wide32.vcf", "-pedFile", TD + "/wide32.ped", "-method", "1", "-v"], tmp_path / "o.vcf")
    assert "Use -method 2" in msg
