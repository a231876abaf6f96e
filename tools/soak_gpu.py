"""tools/soak_gpu.py [first_seed] [n_seeds] [max_members] — GPU soak: random pedigrees (with marriage
loops), random batch sizes around the chunk boundaries, every engine, against the oracle
(tests/_soak.py; seeds 0-39 at max_members 10 are also tests/test_gpu_soak.py).  Prints one line per
seed; exits non-zero on the first mismatch.  Run on the GPU box (compiles kernels there)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _soak import run_seed  # noqa: E402

first, count, max_n = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 0), (2, 20), (3, 10)))
t_start = time.time()
for seed in range(first, first + count):
    line, bad = run_seed(seed, max_n, threads=16)
    print("%s  [%.0f s]" % (line, time.time() - t_start), flush=True)
    if bad:
        sys.exit(1)
