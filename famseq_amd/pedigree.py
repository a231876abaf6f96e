"""PED files and pedigree relations (host side, Python mirror).

Mirrors the reference's PED surface so the parity tests read like it:
  readPed      /root/reference/src/file.cpp:24-62   (skip one header line; stop at
               the first line shorter than 2 chars; columns id mid fid gender name)
  setRelation  /root/reference/src/family.cpp:291-350
  checkPed     /root/reference/src/family.cpp:204-219
The C++ host library (famseq_amd/csrc/host) carries the same logic for the CLI;
this module feeds the ctypes binding, the tests and bench.py.
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np


@dataclass
class Pedigree:
    ids: List[int]
    mids: List[int]
    fids: List[int]
    genders: List[int]
    names: List[str] = field(default_factory=list)

    @property
    def n(self):
        return len(self.ids)

    @property
    def sequenced(self):
        """PED name `NA` = member without a VCF/LK column (file.cpp:207-220)."""
        return np.array([0 if s == "NA" else 1 for s in self.names], dtype=np.uint8)

    def relations(self):
        """-> (mother[], father[]) index arrays, -1 for founders.

        Raises ValueError the way family::init() returns false."""
        n = self.n
        mo = np.full(n, -1, np.int32)
        fa = np.full(n, -1, np.int32)
        for i in range(n):
            im = ifa = -1
            for j in range(n):  # no early exit: the last matching id wins
                if self.mids[i] == self.ids[j]:
                    im = j
                if self.fids[i] == self.ids[j]:
                    ifa = j
            if (im < 0) != (ifa < 0):
                raise ValueError("This is not a fulfill family. Please check the ped file.")
            mo[i], fa[i] = im, ifa
        for i in range(n):
            if mo[i] >= 0:
                if self.genders[mo[i]] != 2:
                    raise ValueError("Sample %d's a mother while she is not a female." % self.ids[mo[i]])
                if self.genders[fa[i]] != 1:
                    raise ValueError("Sample %d's a father while she is not a male." % self.ids[mo[i]])
        return mo, fa


def read_ped(path) -> Pedigree:
    ids, mids, fids, genders, names = [], [], [], [], []
    with open(path) as f:
        f.readline()
        for line in f:
            line = line.rstrip("\n")
            if len(line) < 2:
                break
            t = line.split()
            ids.append(int(t[0]))
            mids.append(int(t[1]))
            fids.append(int(t[2]))
            genders.append(int(t[3]))
            names.append(t[4] if len(t) > 4 else "")
    return Pedigree(ids, mids, fids, genders, names)


def _mk(rows):
    ids, mids, fids, genders = zip(*rows)
    return Pedigree(list(ids), list(mids), list(fids), list(genders), ["s%02d" % i for i in ids])


# SURVEY.md Appendix C — the synthetic benchmark pedigrees (id, mother, father, gender)
PED5 = [(1, 0, 0, 1), (2, 0, 0, 2), (3, 2, 1, 1), (4, 2, 1, 2), (5, 2, 1, 1)]
PED10 = PED5[:4] + [(5, 0, 0, 2), (6, 0, 0, 1), (7, 5, 3, 1), (8, 5, 3, 2), (9, 4, 6, 1), (10, 4, 6, 2)]
PED15 = PED5 + [(6, 0, 0, 2), (7, 0, 0, 1), (8, 0, 0, 2), (9, 6, 3, 1), (10, 6, 3, 2), (11, 4, 7, 1),
                (12, 4, 7, 2), (13, 8, 5, 1), (14, 8, 5, 2), (15, 8, 5, 1)]


def synthetic_pedigree(name) -> Pedigree:
    """ped5 / ped10 / ped15 are BASELINE.json's benchmark pedigrees; trio and quad (the first 3 and 4
    members of ped5) are the shapes most real callers have; sibN = two parents and N - 2 children (tuning aid)."""
    if ":" in name:  # "ped10:8" = the first 8 members of ped10 (tuning aid; the prefix must be closed under parents)
        base, k = name.split(":")
        return _mk({"ped5": PED5, "ped10": PED10, "ped15": PED15}[base][:int(k)])
    if name.startswith("sib"):
        return _mk([(1, 0, 0, 1), (2, 0, 0, 2)] + [(3 + i, 2, 1, 1 + i % 2) for i in range(int(name[3:]) - 2)])
    return _mk({"trio": PED5[:3], "quad": PED5[:4], "ped5": PED5, "ped10": PED10, "ped15": PED15}[name])


def write_ped(ped: Pedigree, path):
    with open(path, "w") as f:
        f.write("ID\tmID\tfID\tgender IndividualName\n")
        for i in range(ped.n):
            f.write("%d\t%d\t%d\t%d\t%s\n" % (ped.ids[i], ped.mids[i], ped.fids[i], ped.genders[i], ped.names[i]))
