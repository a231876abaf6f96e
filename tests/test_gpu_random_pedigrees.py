"""Randomly grown pedigrees (3..9 members; with and without marriage loops, partially sequenced)
through every device engine against the oracle.  Generated kernels are compiled on the box."""
import numpy as np
import pytest

import famseq_amd as fs
import oracle
from famseq_amd.synth import grow_pedigree, random_likelihoods

from famseq_amd.prebuild_sets import RANDOM_SEEDS, random_pedigree

pytestmark = pytest.mark.gpu
RTOL = 1e-9


@pytest.mark.parametrize("seed", RANDOM_SEEDS)
def test_random_pedigree_all_engines(seed):
    rng, ped = random_pedigree(seed)  # famseq_amd/prebuild_sets.py: build() pre-compiles these pedigrees' kernels
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
