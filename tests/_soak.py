"""One seed of the GPU soak: a randomly grown pedigree (3..max_n members, every other seed with
marriage loops, partially sequenced), a batch size around the kernels' chunk boundaries, a mutation
rate from {1e-7, 1e-4, 0}, adversarial likelihoods (hard zeros, shortcut sites) — through every
device engine against the oracle, then the fused call path (packed PLs with missing samples and a
shuffled column order) against the fp64 entry point.  Used by tests/test_gpu_soak.py (a seeded slice
in the suite) and tools/soak_gpu.py (long runs)."""
import math

import numpy as np

import famseq_amd as fs
import oracle
from famseq_amd.prebuild_sets import soak_pedigree  # noqa: F401  (one definition: build() pre-compiles these pedigrees)
from famseq_amd.synth import random_likelihoods

BATCH_SIZES = [1, 63, 64, 255, 256, 257, 511, 513, 1000, 1279, 2048]


def run_seed(seed, max_n=10, threads=8):
    """-> (summary line, [failure descriptions])."""
    rng, ped, mu = soak_pedigree(seed, max_n)
    n = ped.n
    n_sites = int(rng.choice(BATCH_SIZES))
    if 3 ** n * n_sites > 4e8:
        n_sites = max(1, int(4e8 / 3 ** n))
    lk, flags = random_likelihoods(rng, ped, n_sites)
    ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders, ped.sequenced, mrate=mu).bn_batch(lk, flags, threads=threads)
    model = fs.make_model(ped, mrate=mu)
    probe = fs.Context(model, device=-1)
    plan = probe.plan()
    probe.close()
    engines = [("team", dict(enum_impl=0)), ("lane", dict(enum_impl=1))]
    if plan["enum_group_digits_max"] >= 1:  # lanes-per-site mode, 3 or 9 lanes per site
        engines.append(("group", dict(enum_impl=1, group_digits=min(2, plan["enum_group_digits_max"]))))
    if plan["elim_supported"]:
        engines.append(("elim", dict(engine=fs.ENGINE_ELIM)))
    res, bad = [], []
    for name, opt in engines:
        ctx = fs.Context(model, **opt)
        post, single, st = ctx.bn_batch(lk, flags)
        ctx.close()
        ok, s_ok = (ref[2] & 3) == 0, (ref[2] & 3) != 1
        good = np.array_equal(st, ref[2]) and np.array_equal(single[s_ok], ref[1][s_ok]) and \
            np.allclose(post[ok], ref[0][ok], rtol=1e-9, atol=0) and bool(np.all(np.isnan(post[~ok])))
        res.append("%s %s" % (name, "ok" if good else "MISMATCH"))
        if not good:
            bad.append(name)
    # the fused call path on the same pedigree: packed integer PLs (with missing samples, a PL beyond
    # the table, a shuffled column order) against the fp64 entry point fed with the host's table
    seq = np.nonzero(ped.sequenced)[0].astype(np.int32)
    rng.shuffle(seq)
    k = len(seq)
    pl = rng.randint(0, 400, size=(n_sites, k, 3)).astype(np.uint16)
    pl[np.arange(n_sites)[:, None], np.arange(k)[None, :], rng.randint(0, 3, size=(n_sites, k))] = 0
    pl[rng.rand(n_sites, k) < 0.05] = fs.PL_MISSING
    pl[rng.rand(n_sites, k, 3) < 0.01] = 5000
    table = np.zeros(4097)
    table[:4096] = [math.pow(10.0, -i / 10.0) for i in range(4096)]
    lk2 = np.ones((n_sites, n, 3))
    for j, mbr in enumerate(seq):
        v = table[np.minimum(pl[:, j].astype(np.int64), 4096)]
        v[np.all(pl[:, j] == fs.PL_MISSING, axis=1)] = 1.0
        lk2[:, mbr] = v
    for name, opt in engines[1:]:
        if name == "group":
            continue
        ctx = fs.Context(model, **opt)
        a = ctx.bn_call_batch(seq, lk=lk2, flags=flags)
        b = ctx.bn_call_batch(seq, pl16=pl, flags=flags)
        post2, single2, st2 = ctx.bn_batch(lk2, flags)
        text, st3 = ctx.bn_call_text_batch(seq, pl16=pl, flags=flags)
        ctx.close()
        good = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b)) and np.array_equal(a[3], st2) and np.array_equal(st3, st2)
        # the text records of a sample of (site, column) pairs: Python's '%g' of the numbers (what printf / ostream print)
        computed = np.nonzero((st2 & 3) == 0)[0]
        for s_ in (rng.choice(computed, size=min(40, len(computed)), replace=False) if len(computed) else []):
            for j in range(k):
                rec = text[s_, j]
                want = "%g,%g,%g:%g,%g,%g:%s\t" % (tuple(a[0][s_, j]) + tuple(a[1][s_, j]) + ({0: "0/0", 1: "0/1"}.get(int(a[2][s_, j]), "1/1"),))
                good = good and bytes(rec[:rec[-1]]) == want.encode()
        okc = (st2 & 3) == 0
        good = good and np.array_equal(a[2][okc], fs.call_genotypes(post2[okc][:, seq]).reshape(-1, k))
        res.append("call/%s %s" % (name, "ok" if good else "MISMATCH"))
        if not good:
            bad.append("call/" + name)
    line = "seed %3d n=%2d sites=%5d cond=%d  %s" % (seed, n, n_sites, plan["elim_conditioned_members"], ", ".join(res))
    return line, bad
