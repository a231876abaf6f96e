"""Randomly grown pedigrees (3..9 members; with and without marriage loops, partially sequenced)
through every device engine against the oracle.  Generated kernels are compiled on the box."""
import numpy as np
import pytest

import famseq_amd as fs
import oracle

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def grow_pedigree(rng, n, allow_loops):
    """Start from founders, keep adding children of random (female, male) couples; without
    allow_loops a couple is only formed if it does not close a loop in the member/family graph."""
    ids, mids, fids, gen = [1, 2], [0, 0], [0, 0], [1, 2]
    comp = {1: 1, 2: 2}  # connected component of each member (for loop avoidance)
    couples = {}
    while len(ids) < n:
        r = rng.rand()
        females = [i for i, g in zip(ids, gen) if g == 2]
        males = [i for i, g in zip(ids, gen) if g == 1]
        if r < 0.3 or not females or not males:
            new = len(ids) + 1
            ids.append(new); mids.append(0); fids.append(0); gen.append(int(rng.randint(1, 3)))
            comp[new] = new
            continue
        mo, fa = int(rng.choice(females)), int(rng.choice(males))
        if (mo, fa) not in couples:
            if not allow_loops and comp[mo] == comp[fa]:
                continue
            couples[(mo, fa)] = True
            old = comp[fa]
            for k in comp:
                if comp[k] == old:
                    comp[k] = comp[mo]
        new = len(ids) + 1
        ids.append(new); mids.append(mo); fids.append(fa); gen.append(int(rng.randint(1, 3)))
        comp[new] = comp[mo]
    names = ["s%d" % i if rng.rand() < 0.75 else "NA" for i in ids]
    if all(x == "NA" for x in names):
        names[-1] = "s_last"
    return fs.Pedigree(ids, mids, fids, gen, names)


def random_likelihoods(rng, ped, n_sites):
    pl = rng.randint(0, 300, size=(n_sites, ped.n, 3)).astype(float)
    pl[np.arange(n_sites)[:, None], np.arange(ped.n)[None, :], rng.randint(0, 3, size=(n_sites, ped.n))] = 0
    lk = 10.0 ** (-pl / 10.0)
    lk[rng.rand(n_sites, ped.n, 3) < 0.02] = 0.0           # hard zeros
    sharp = rng.rand(n_sites) < 0.15                        # some sites take the -LRC shortcut
    lk[sharp] = np.where(pl[sharp] == 0, 1.0, 1e-40)
    lk[:, ped.sequenced == 0, :] = 1.0
    return lk, rng.randint(0, 4, n_sites).astype(np.uint8)


@pytest.mark.parametrize("seed", range(10))
def test_random_pedigree_all_engines(seed):
    rng = np.random.RandomState(1000 + seed)
    n = int(rng.randint(3, 10))
    ped = grow_pedigree(rng, n, allow_loops=seed % 3 == 0)
    ped.relations()
    mu = [1e-7, 1e-7, 1e-4, 0.0][seed % 4]
    lk, flags = random_likelihoods(rng, ped, 96)
    ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders, ped.sequenced, mrate=mu).bn_batch(lk, flags, threads=4)
    assert len(np.unique(ref[2])) >= 2  # the mix of sites exercises several statuses
    model = fs.make_model(ped, mrate=mu)
    probe = fs.Context(model, device=-1)
    engines = [dict(enum_impl=0), dict(enum_impl=1)]
    if probe.plan()["elim_supported"]:
        engines.append(dict(engine=fs.ENGINE_ELIM))
    probe.close()
    for opt in engines:
        ctx = fs.Context(model, **opt)
        post, single, st = ctx.bn_batch(lk, flags)
        ctx.close()
        assert np.array_equal(st, ref[2]), opt
        ok, s_ok = (st & 3) == 0, (st & 3) != 1
        assert np.array_equal(single[s_ok].view(np.uint64), ref[1][s_ok].view(np.uint64)), opt
        np.testing.assert_allclose(post[ok], ref[0][ok], rtol=RTOL, atol=0, err_msg=str(opt))
        assert np.all(np.isnan(post[~ok]))
