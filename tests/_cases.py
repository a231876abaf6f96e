"""Shared helpers: load tests/golden/*.npz into (pedigree, constants, arrays) cases."""
import os

import numpy as np

from famseq_amd.pedigree import Pedigree

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Case:
    def __init__(self, name, z, key):
        g = lambda f: z[key + "." + f]
        ped = g("ped")
        self.name = name
        self.ids, self.mids, self.fids, self.genders = (ped[i].tolist() for i in range(4))
        self.sequenced = g("sequenced")
        self.lk, self.flags = g("lk"), g("flags")
        self.post, self.single, self.status = g("post"), g("single"), g("status")
        self.peel = z[key + ".peel"] if key + ".peel" in z else None
        self.consts = {}
        for src, dst in (("mrate", "mrate"), ("lc", "lc"), ("gN", "genoProbN"), ("gK", "genoProbK"),
                         ("gXN", "genoProbXN"), ("gXK", "genoProbXK")):
            if key + "." + src in z:
                v = z[key + "." + src]
                self.consts[dst] = float(v) if v.ndim == 0 else v.tolist()

    @property
    def n(self):
        return len(self.ids)

    def pedigree(self):
        names = ["s%d" % i if s else "NA" for i, s in zip(self.ids, self.sequenced)]
        return Pedigree(self.ids, self.mids, self.fids, self.genders, names)

    def __repr__(self):
        return self.name


def load_cases(files=("bn_vcf.npz", "bn_lk.npz", "bn_synth.npz")):
    out = []
    for fn in files:
        z = np.load(os.path.join(GOLDEN, fn))
        keys = sorted({k.rsplit(".", 1)[0] for k in z.files if k.endswith(".lk")})
        for k in keys:
            out.append(Case("%s:%s" % (fn.replace(".npz", ""), k), z, k))
    return out
