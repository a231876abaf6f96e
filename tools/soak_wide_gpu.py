"""tools/soak_wide_gpu.py [first_seed] [n_seeds] [max_members] — GPU soak of the sum-product engine beyond 20 members: random
loop-free pedigrees of 21..max_members members (partially sequenced), adversarial rows, mu in {1e-7, 1e-4, 0}, ragged batch
sizes, against the numpy sum-product oracle (pinned to the compiled reference's -method 2) on the sites where its double
arithmetic reproduces its own long-double result (tests/test_gpu_wide.py explains why).  One line per seed; exits non-zero on
the first mismatch.  Run on the GPU box (compiles kernels there)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import famseq_amd as fs  # noqa: E402
import oracle.sum_product as sp  # noqa: E402

first, count, max_n = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 0), (2, 20), (3, 80)))
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.RandomState(77000 + seed)
    ped = fs.synth.grow_pedigree(rng, int(rng.randint(21, max_n + 1)), allow_loops=False)
    mu = [1e-7, 1e-4, 0.0][seed % 3]
    s = int(rng.choice([1, 63, 64, 65, 255, 257, 700]))
    lk, flags = fs.synth.random_likelihoods(rng, ped, s, max_pl=60)
    want = sp.pedigree_posterior(ped, lk, flags, mrate=mu)
    true = sp.pedigree_posterior(ped, lk, flags, mrate=mu, dtype=np.longdouble)
    ctx = fs.Context(fs.make_model(ped, mrate=mu), device=0)
    variant = ctx.plan()["elim_variant"]
    post, single, status = ctx.bn_batch(lk, flags)
    ctx.close()
    sound = (want[2] == true[2]) & np.all(np.isclose(want[0], true[0].astype(np.float64), rtol=1e-10, atol=1e-35, equal_nan=True), axis=(1, 2))
    ok, s_ok = sound & ((want[2] & 3) == 0), (want[2] & 3) != 1
    loose = ~sound & (status == 0)
    good = (np.array_equal(status[sound], want[2][sound]) and np.array_equal(single[s_ok], want[1][s_ok]) and
            np.allclose(post[ok], want[0][ok], rtol=1e-9, atol=1e-35) and bool(np.all(np.isin(status[~sound], (0, 2)))) and
            bool(np.all(np.isfinite(post[loose]))) and bool(np.all(np.abs(post[loose].sum(axis=2) - 1) < 1e-6)) and
            bool(np.all(np.isnan(post[(status & 3) != 0]))))
    print("seed %3d n=%3d sites=%4d variant=%2d sound=%.3f %s  [%.0f s]" % (seed, ped.n, s, variant, sound.mean(), "ok" if good else "MISMATCH", time.time() - t0), flush=True)
    if not good:
        sys.exit(1)
