"""Fixtures of the wide pedigrees (more than 20 members): tests/golden/wide_peds.npz, written by
oracle/gen_golden_wide.py from the compiled reference's -method 2 (family::calPostProbPeeling)."""
import os

import numpy as np

from famseq_amd.pedigree import Pedigree

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIZES = (24, 32, 48)


class WideCase:
    def __init__(self, n):
        d = np.load(os.path.join(GOLDEN, "wide_peds.npz"))
        k = "wide%d" % n
        self.name = k
        self.ped = Pedigree([int(x) for x in d[k + "_ids"]], [int(x) for x in d[k + "_mids"]], [int(x) for x in d[k + "_fids"]],
                            [int(x) for x in d[k + "_genders"]], [str(x) for x in d[k + "_names"]])
        self.lk, self.flags = d[k + "_lk"], d[k + "_flags"]
        self.post, self.single, self.status = d[k + "_post"], d[k + "_single"], d[k + "_status"]


def check(case, post, single, status, rtol=1e-9):
    """status identical, single posterior bit-exact, posterior within rtol (exact zeros equal), failed rows NaN."""
    assert np.array_equal(status, case.status)
    ok, s_ok = (case.status & 3) == 0, (case.status & 3) != 1
    assert np.array_equal(single[s_ok], case.single[s_ok])
    np.testing.assert_allclose(post[ok], case.post[ok], rtol=rtol, atol=0)
    assert np.all(np.isnan(post[~ok]))
    assert np.all(np.isnan(single[~s_ok]))
