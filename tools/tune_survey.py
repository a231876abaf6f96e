"""tools/tune_survey.py [first_seed] [n_seeds] [max_members] — what famseq_set_option(ctx, "tune", 1) picks on randomly grown
pedigrees (tests/_soak.py's generator), one line per pedigree: members, nuclear families, conditioned members, the report.
Data for the static rules that choose when nothing has been tuned (elim_first_variant, the 7-member block)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("FAMSEQ_KERNEL_CACHE", tempfile.mkdtemp(prefix="famseq_survey_"))
import famseq_amd as fs  # noqa: E402
from _soak import soak_pedigree  # noqa: E402

first, count, max_n = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 5000), (2, 40), (3, 14)))
t0 = time.time()
for seed in range(first, first + count):
    _, ped, mu = soak_pedigree(seed, max_n)
    if ped.n < 8:
        continue
    ctx = fs.Context(fs.make_model(ped, mrate=mu))
    try:
        ctx.set_option("tune", 1)
        p = ctx.plan()
        fams = len({(m, f) for m, f in zip(ped.mids, ped.fids) if m})
        print("seed %d n=%d families=%d cond=%d | %s  [%.0f s]" % (seed, ped.n, fams, p["elim_conditioned_members"], p["tune"], time.time() - t0), flush=True)
    except fs.FamseqError as e:
        print("seed %d n=%d: %s" % (seed, ped.n, e), flush=True)
    ctx.close()
