"""tools/io_kernel_rates.py — run the fused call path (packed PLs in, Phred + genotype call out) on
seeded ped10 sites so that `rocprofv3 --kernel-trace --stats -- python3 tools/io_kernel_rates.py`
shows unpack_pl16_kernel / phred_call_kernel per-launch durations next to the posterior kernel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import famseq_amd as fs

ped = fs.synthetic_pedigree(sys.argv[3] if len(sys.argv) > 3 else "ped10")
n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.RandomState(1)
pl = rng.randint(0, 400, size=(n_sites, ped.n, 3)).astype(np.uint16)
pl[np.arange(n_sites), :, rng.randint(0, 3, n_sites)] = 0
ctx = fs.Context(fs.make_model(ped))
if not (len(sys.argv) > 2 and sys.argv[2] == "enum"):  # default: the sum-product engine's call-path form
    ctx.set_option("engine", fs.ENGINE_ELIM)
ctx.set_option("chunk_sites", 1 << 20)
seq = np.arange(ped.n, dtype=np.int32)
for _ in range(3):
    gpp, fpp, fgt, st = ctx.bn_call_batch(seq, pl16=pl)
print("ok", n_sites, int((st & 3 != 0).sum()))
print("bytes per site: unpack %d in + %d out; phred %d in + %d out" % (6 * ped.n, 24 * ped.n, 48 * ped.n + 1, 49 * ped.n))
