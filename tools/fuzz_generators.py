"""tools/fuzz_generators.py [first_seed] [n_seeds] [max_members] [loops] — host-side fuzz of the two kernel
generators: random pedigrees (tests/test_gpu_random_pedigrees.grow_pedigree) -> generated source for a
one-lane workgroup -> g++ -> compare with the oracle (the machinery of tests/test_generated_host.py).
No GPU involved; prints one line per seed and exits non-zero on the first mismatch."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import famseq_amd as fs  # noqa: E402
import oracle  # noqa: E402
from test_generated_host import build_host_kernel, run_host  # noqa: E402
from test_gpu_random_pedigrees import grow_pedigree, random_likelihoods  # noqa: E402


class Env:  # the little of pytest's monkeypatch that build_host_kernel uses
    def setenv(self, k, v):
        os.environ[k] = v


first, count, max_n = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 0), (2, 20), (3, 11)))
always_loops = len(sys.argv) > 4 and sys.argv[4] == "loops"  # every pedigree may close marriage loops
bad = 0
for seed in range(first, first + count):
    rng = np.random.RandomState(5000 + seed)
    n = int(rng.randint(3, max_n + 1))
    ped = grow_pedigree(rng, n, allow_loops=always_loops or seed % 4 == 0)
    ped.relations()
    mu = [1e-7, 1e-4, 0.0][seed % 3]
    lk, flags = random_likelihoods(rng, ped, 24 if n > 9 else 48)
    ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders, ped.sequenced, mrate=mu).bn_batch(lk, flags, threads=4)
    model = fs.make_model(ped, mrate=mu)
    probe = fs.Context(model, device=-1)
    plan = probe.plan()
    probe.close()
    for kind in ["lane"] + (["elim"] if plan["elim_supported"] else []):
        with tempfile.TemporaryDirectory() as d:
            import pathlib
            fn = build_host_kernel(model, kind, pathlib.Path(d), Env())
            post, single, st = run_host(fn, model, lk, flags)
        ok, s_ok = (ref[2] & 3) == 0, (ref[2] & 3) != 1
        good = np.array_equal(st, ref[2]) and np.array_equal(single[s_ok], ref[1][s_ok]) and \
            np.allclose(post[ok], ref[0][ok], rtol=1e-10, atol=0)
        print("seed %3d n=%2d %-4s %s  (%s)" % (seed, n, kind, "ok" if good else "MISMATCH", plan["enum_lane_shape"] if kind == "lane" else "statuses %s, %d conditioned" % (sorted(set(st.tolist())), plan["elim_conditioned_members"])), flush=True)
    if not plan["elim_supported"]:
        print("seed %3d n=%2d elim not supported (more than three conditioning members)" % (seed, n), flush=True)
        bad += not good
sys.exit(1 if bad else 0)
