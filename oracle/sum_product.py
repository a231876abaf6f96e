"""oracle/sum_product.py — TEST INFRASTRUCTURE ONLY (never imported by famseq_amd/, see oracle/__init__.py).

CPU restatement, in numpy, of what the reference's `-method 2` returns for a LOOP-FREE pedigree of any size:
`family::calPostProbPeeling` (/root/reference/src/family.cpp:1126-1403 driver, :1501-1845 the recursive
anterior / posterior terms).  The driver is the one `calPostProbBN` has (single posterior :1134, shortcut vote
:1140-1163, shortcut branch :1169-1260, row normalisation with the `sum <= 0 -> false` rule); the peeling itself
computes, for every member, the exact marginal of the pedigree's Bayesian network — the quantity the 3^N
enumeration of `-method 1` (family.cpp:882-954) sums out.  On a forest those marginals are given by two-way
message passing over the member / nuclear-family graph, which is what this file does (the reference's anterior
term of a member is the message from its parents' family, its posterior terms are the messages from the families
it is a parent of); it does NOT follow the reference's recursion order, so results agree to rounding (pinned to the
compiled reference's `calPostProbPeeling` at 1e-9 relative on tests/golden/wide_peds.npz: tests/test_oracle_golden.py),
not bit for bit.  The enumeration oracle (bn_oracle.c) cannot check pedigrees of more than 20 members — 3^24 is
2.8e11 configurations per site — which is what this one is for.

Vectorised over sites; sites are grouped by their flags byte (bit0 Known picks the founders' priors, bit1 chrX the
transmission tables and the male founders' priors: family.cpp:990-1073).
"""
import numpy as np

ST_SINGLE_FAIL, ST_BN_FAIL, ST_SHORTCUT = 1, 2, 0x80


def _families(mother, father):
    fam = {}
    for i, (mo, fa) in enumerate(zip(mother, father)):
        if mo >= 0:
            fam.setdefault((int(mo), int(fa)), []).append(i)
    return fam


def is_forest(mother, father):
    """True when the member / nuclear-family graph has no cycle (the domain of -method 2)."""
    n = len(mother)
    fam = _families(mother, father)
    parent = list(range(n + len(fam)))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for k, ((mo, fa), kids) in enumerate(fam.items()):
        for p in [mo, fa] + kids:
            a, b = find(p), find(n + k)
            if a == b:
                return False
            parent[a] = b
    return True


def posterior(mother, father, gender, sequenced, pcp2, pcp2Xf, pcp2Xm, gN, gK, gXN, gXK, lc, lk, flags=None, dtype=np.float64):
    """-> (post, single, status) as famseq_bn_batch returns them.  mother / father: index or -1; tables are
    27 doubles [child*9 + mother*3 + father] (family.cpp:447-550, :383-445).
    dtype=np.longdouble evaluates the same messages with x87's 15-bit exponent: products that underflow in double —
    where the reference's peeling returns false (status 2) — come out, which tells an underflow from a real zero."""
    mother, father, gender = (np.asarray(a, dtype=np.int64) for a in (mother, father, gender))
    n = len(mother)
    lk = np.ascontiguousarray(lk, dtype=dtype).reshape(-1, n, 3)
    s_all = lk.shape[0]
    flags = np.zeros(s_all, np.uint8) if flags is None else np.asarray(flags, dtype=np.uint8)
    seq = np.ones(n, bool) if sequenced is None else np.asarray(sequenced).astype(bool)
    post = np.full((s_all, n, 3), np.nan, dtype=dtype)
    single = np.full((s_all, n, 3), np.nan, dtype=dtype)
    status = np.zeros(s_all, np.uint8)
    fam = _families(mother, father)
    fam_keys = list(fam)
    adjacent = [[] for _ in range(n)]
    for k, (mo, fa) in enumerate(fam_keys):
        for p in [mo, fa] + fam[(mo, fa)]:
            adjacent[p].append(k)
    for fl in range(4):
        idx = np.nonzero((flags & 3) == fl)[0]
        if idx.size == 0:
            continue
        known, chrx = fl & 1, fl & 2
        autos = np.asarray(gK if known else gN, dtype=dtype)
        male = np.asarray(gXK if known else gXN, dtype=dtype) if chrx else autos
        prior = [male if gender[p] == 1 else autos for p in range(n)]  # family.cpp:1052-1062, :1410-1424
        table = [np.asarray((pcp2Xm if gender[p] == 1 else pcp2Xf) if chrx else pcp2, dtype=dtype).reshape(3, 3, 3)
                 for p in range(n)]  # [child][mother][father]; family.cpp:1063-1073
        l = lk[idx]
        # calPostProbSingle, family.cpp:1405-1499
        sp = l * np.stack(prior)[None]
        ssum = (sp[:, :, 0] + sp[:, :, 1]) + sp[:, :, 2]
        s_fail = (ssum <= 0).any(axis=1)
        with np.errstate(invalid="ignore", divide="ignore"):
            sg = sp / ssum[:, :, None]
        # shortcut vote over the sequenced members, family.cpp:1140-1163
        with np.errstate(invalid="ignore", divide="ignore"):
            big = l.max(axis=2) / ((l[:, :, 0] + l[:, :, 1]) + l[:, :, 2])
        full = ((big < lc) & seq[None]).any(axis=1)
        # message passing
        local = [l[:, p, :] * (prior[p][None] if mother[p] < 0 else 1.0) for p in range(n)]
        memo = {}

        def v2f(p, k):
            key = ("v", p, k)
            if key not in memo:
                m = local[p]
                for k2 in adjacent[p]:
                    if k2 != k:
                        m = m * f2v(k2, p)
                memo[key] = m
            return memo[key]

        def f2v(k, t):
            key = ("f", k, t)
            if key in memo:
                return memo[key]
            mo, fa = fam_keys[k]
            c = np.ones((l.shape[0], 3, 3), dtype=dtype)  # [site][gm][gf]
            if t != mo:
                c = c * v2f(mo, k)[:, :, None]
            if t != fa:
                c = c * v2f(fa, k)[:, None, :]
            for kid in fam[(mo, fa)]:
                if kid != t:
                    c = c * np.einsum("cmf,sc->smf", table[kid], v2f(kid, k))
            if t == mo:
                out = c.sum(axis=2)
            elif t == fa:
                out = c.sum(axis=1)
            else:
                out = np.einsum("cmf,smf->sc", table[t], c)
            memo[key] = out
            return out

        bn = np.empty_like(l)
        for p in range(n):
            m = local[p]
            for k in adjacent[p]:
                m = m * f2v(k, p)
            bn[:, p, :] = m
        bsum = (bn[:, :, 0] + bn[:, :, 1]) + bn[:, :, 2]
        b_fail = (bsum <= 0).any(axis=1)
        with np.errstate(invalid="ignore", divide="ignore"):
            bn = bn / bsum[:, :, None]
        st = np.where(s_fail, ST_SINGLE_FAIL, np.where(~full, ST_SHORTCUT, np.where(b_fail, ST_BN_FAIL, 0))).astype(np.uint8)
        p_out = np.where((st == 0)[:, None, None], bn, np.where((st == ST_SHORTCUT)[:, None, None], sg, np.nan))
        s_out = np.where((st == ST_SINGLE_FAIL)[:, None, None], np.nan, sg)
        post[idx], single[idx], status[idx] = p_out, s_out, st
    return post, single, status


def pedigree_posterior(ped, lk, flags=None, mrate=1e-7, lc=1.0, gN=(0.9985, 0.001, 0.0005), gK=(0.45, 0.1, 0.45),
                       gXN=(0.999, 0, 0.001), gXK=(0.5, 0, 0.5), dtype=np.float64):
    """`posterior` for a famseq_amd.pedigree.Pedigree with the reference's default priors (family.cpp:91-109) and the
    tables of oracle.tables(mrate) (bn_oracle.c, pinned bit-exact to the reference's calPCP2S / calPCP2Xf / calPCP2Xm)."""
    import oracle

    mo, fa = ped.relations()
    t = oracle.tables(mrate)
    return posterior(mo, fa, ped.genders, ped.sequenced, t[0], t[1], t[2], gN, gK, gXN, gXK, lc, lk, flags, dtype)
