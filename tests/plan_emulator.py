"""A numpy walk of the enumeration PLAN (test helper, small pedigrees only).

This is not a compute path of the product: it replays, lane by lane and step by step,
exactly the index arithmetic the HIP kernel performs with the plan tables exported by
famseq_plan_json (laneoff / joff / jdigits / bin descriptors), so the host-side plan
builder can be checked against the oracle without a GPU.  It checks that the tiling
visits every one of the 3^N configurations once and bins it correctly."""
import itertools

import numpy as np


def factor_tables(model):
    """Tc[fl][kind][27] as bn_kernel.hip::build_factor_tables."""
    tc = np.zeros((4, 4, 27))
    gN, gK = np.array(model.genoProbN[:]), np.array(model.genoProbK[:])
    gXN, gXK = np.array(model.genoProbXN[:]), np.array(model.genoProbXK[:])
    p, xf, xm = (np.array(getattr(model, k)[:]) for k in ("pcp2", "pcp2Xf", "pcp2Xm"))
    for fl in range(4):
        known, x = fl & 1, fl & 2
        autos = gK if known else gN
        male = (gXK if known else gXN) if x else autos
        for g in range(3):
            tc[fl, 0, 9 * g] = male[g]
            tc[fl, 1, 9 * g] = autos[g]
        tc[fl, 2] = xm if x else p
        tc[fl, 3] = xf if x else p
    return tc.reshape(4, 108)


def emulate_site(plan, model, lk, fl):
    """-> unnormalised marginals [N,3] and the number of configurations visited."""
    N, L, A, J = plan["N"], plan["L"], plan["A"], plan["J"]
    TL, ns, tab = plan["team_lanes"], plan["n_slots"], plan["iter_tab"]
    nA, nB, stride, ss = plan["nA"], plan["nB"], plan["row_stride"], plan["step_slots"]
    assert nA + L + nB == ns and L + nB <= ss and stride >= ns and stride % 2 == 1
    laneoff = np.array(plan["laneoff"], dtype=np.int64).reshape(TL, stride)
    joff = np.array(plan["joff"], dtype=np.int64).reshape(3, tab, ss)
    jdig = np.array(plan["jdigits"], dtype=np.int64).reshape(3, tab)
    tcf = factor_tables(model)[fl]
    lkf = lk.reshape(-1)

    def factor(pk, g9=0, g1=0):  # packed BYTE offsets -> table entry * likelihood entry
        assert (pk & 0xFFFF) % 8 == 0 and (pk >> 16) % 8 == 0
        return tcf[(pk & 0xFFFF) // 8 + g9] * lkf[(pk >> 16) // 8 + g1]

    cols = plan["cols"]
    red = np.zeros((cols, TL))
    visited = 0
    jn, jd = plan["jn"], plan["jd"]
    low_dep = False
    for t in range(TL):
        row = laneoff[t]
        pA = 1e7
        for s in range(nA):
            pA = pA * factor(row[s])
        for j2, j1, j0 in itertools.product(range(jn[2]), range(jn[1]), range(jn[0])):
            rec = joff[0, j0] + joff[1, j1] + joff[2, j2]
            low_dep = low_dep or bool(np.any(rec[:L]))
            pj = pA
            for s in range(nB):
                pj = pj * factor(row[nA + L + s] + rec[L + s])
            v = np.zeros((L, 3))
            for k in range(L):
                pk = row[nA + k] + rec[k]
                for g in range(3):
                    v[k, g] = factor(pk, 9 * g, g)
            tot = 0.0
            for cfg in itertools.product(range(3), repeat=L):
                w = pj
                for k in range(L):
                    w = w * v[k, cfg[k]]
                for k in range(L):
                    red[3 * k + cfg[k], t] += w
                tot += w
                visited += 1
            red[cols - 1, t] += tot
            for lvl, jl in ((0, j0), (1, j1), (2, j2)):
                for d in range(jd[lvl]):
                    g = (jdig[lvl, jl] >> (2 * d)) & 3
                    red[3 * L + 3 * (5 * lvl + d) + g, t] += tot
    assert not (plan["low_invariant"] and low_dep), "plan claims low factors are step-invariant"
    bins = np.zeros((N, 3))
    for i in range(N):
        bk, bi = plan["bin_kind"][i], plan["bin_index"][i]
        for g in range(3):
            if bk == 2:
                sel = [t for t in range(TL) if (t // 3 ** bi) % 3 == g]
                bins[i, g] = red[cols - 1, sel].sum()
            else:
                bins[i, g] = red[(3 * bi if bk == 0 else 3 * L + 3 * bi) + g].sum()
    return bins, visited
