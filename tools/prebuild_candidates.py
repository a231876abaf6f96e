"""tools/prebuild_candidates.py — here (no GPU): compile every candidate famseq_set_option "tune" races — the enumeration
kernel's 7- and 6-member blocks, the sum-product kernel's four starting variants — for the pedigrees
tools/make_tuned_picks.py measures, into the in-tree cache, so that the GPU box spends its minutes timing, not compiling."""
import os
import sys
from concurrent.futures import ProcessPoolExecutor
import multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(rows):
    import famseq_amd as fs
    from famseq_amd.pedigree import Pedigree

    ped = Pedigree(*rows)
    ctx = fs.Context(fs.make_model(ped), device=-1)
    for v in (0, 2):
        ctx.set_option("prebuild_lane", v)
    if ctx.plan()["elim_supported"]:
        for v in (0, 1, 4, 5):
            ctx.set_option("prebuild_elim", v)
    ctx.close()
    return ped.n


if __name__ == "__main__":
    import __graft_entry__ as ge

    peds = ge.tuned_pedigrees()
    rows = [(p.ids, p.mids, p.fids, p.genders, p.names) for p in peds]
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1), mp_context=mp.get_context("spawn")) as ex:
        print(len(list(ex.map(one, rows))), "pedigrees: candidates compiled")
