"""Seeded synthetic sites for the benchmark pedigrees.

One SplitMix64 stream, seed ``0xFA5E0000 + config#``.  Every site consumes a
fixed number of draws D = 2 + 4N so that draw k of site s is stream element
s*D + k (counter-based => vectorisable and shardable: rank r generates its own
site range without touching the others):

  k = 0           allele frequency q = 0.01 + 0.49*u
  k = 1 + 2i, +1  member i, PED order: founders draw two alleles Bernoulli(q);
                  children inherit one allele from each parent (a het parent
                  transmits either allele with probability 1/2; no mutation)
  k = 2N + 1      Known (ID = rs<site>) with probability 0.10
  k = 2N+2+2i,+1  member i: integer PLs around the true genotype t:
                  PL[t] = 0; a = 3 + r % 88 for the adjacent genotype(s); the far
                  genotype gets min(255, a + 10 + r' % 156); for t = 1 both
                  neighbours get independent a.
  u = (z >> 11) * 2^-53,  r = z >> 33.
Likelihood = 10^(-PL/10) (the reference's PL transform, file.cpp:589).  No site
can take the -LRC shortcut (that needs both non-zero PLs >= 160).
"""
import math

import numpy as np

from .pedigree import Pedigree

GAMMA = np.uint64(0x9E3779B97F4A7C15)
M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)
SEED_BASE = 0xFA5E0000

# 10^(-k/10) through libm pow, k = 0..255
PL_LUT = np.array([math.pow(10.0, -k / 10.0) for k in range(256)], dtype=np.float64)


def splitmix64_at(seed, index):
    """Stream element `index` (0-based) of SplitMix64(seed); `index` is a uint64 array."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (index + np.uint64(1)) * GAMMA
        z = (z ^ (z >> np.uint64(30))) * M1
        z = (z ^ (z >> np.uint64(27))) * M2
        return z ^ (z >> np.uint64(31))


def _u01(z):
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def gen_sites(mother, father, n_sites, seed, first_site=0, chunk=1 << 18):
    """-> (pl[int16 S,N,3], known[bool S], true_geno[int8 S,N]) for sites
    [first_site, first_site + n_sites)."""
    mother = np.asarray(mother)
    father = np.asarray(father)
    n = len(mother)
    D = 2 + 4 * n
    order = _topo(mother, father)
    pl = np.empty((n_sites, n, 3), np.int16)
    known = np.empty(n_sites, bool)
    geno = np.empty((n_sites, n), np.int8)
    for lo in range(0, n_sites, chunk):
        hi = min(n_sites, lo + chunk)
        site = np.arange(first_site + lo, first_site + hi, dtype=np.uint64)
        base = site * np.uint64(D)

        def draw(k):
            return splitmix64_at(seed, base + np.uint64(k))

        q = 0.01 + 0.49 * _u01(draw(0))
        g = np.zeros((hi - lo, n), np.int8)
        for i in order:
            u1, u2 = _u01(draw(1 + 2 * i)), _u01(draw(2 + 2 * i))
            if mother[i] < 0:
                g[:, i] = (u1 < q).astype(np.int8) + (u2 < q).astype(np.int8)
            else:
                gm, gf = g[:, mother[i]], g[:, father[i]]
                am = np.where(gm == 1, (u1 < 0.5), gm == 2)
                af = np.where(gf == 1, (u2 < 0.5), gf == 2)
                g[:, i] = am.astype(np.int8) + af.astype(np.int8)
        known[lo:hi] = _u01(draw(2 * n + 1)) < 0.10
        for i in range(n):
            r1 = (draw(2 * n + 2 + 2 * i) >> np.uint64(33)).astype(np.int64)
            r2 = (draw(2 * n + 3 + 2 * i) >> np.uint64(33)).astype(np.int64)
            a1 = 3 + r1 % 88
            a2 = 3 + r2 % 88
            far = np.minimum(255, a1 + 10 + r2 % 156)
            t = g[:, i]
            p0 = np.where(t == 0, 0, np.where(t == 1, a1, far))
            p1 = np.where(t == 1, 0, a1)
            p2 = np.where(t == 2, 0, np.where(t == 1, a2, far))
            pl[lo:hi, i, 0], pl[lo:hi, i, 1], pl[lo:hi, i, 2] = p0, p1, p2
        geno[lo:hi] = g
    return pl, known, geno


def _topo(mother, father):
    n = len(mother)
    done, order = [False] * n, []
    while len(order) < n:
        progressed = False
        for i in range(n):
            if not done[i] and (mother[i] < 0 or (done[mother[i]] and done[father[i]])):
                done[i] = True
                order.append(i)
                progressed = True
        if not progressed:
            raise ValueError("pedigree has a cycle")
    return order


def pl_to_lk(pl):
    """Integer PL 0..255 -> likelihood via the libm-pow lookup table."""
    return PL_LUT[np.asarray(pl, dtype=np.int64)]


def gen_batch(mother, father, n_sites, config_no, first_site=0):
    """-> (lk[float64 S,N,3], flags[uint8 S]) ready for famseq_bn_batch."""
    pl, known, _ = gen_sites(mother, father, n_sites, SEED_BASE + config_no, first_site)
    return pl_to_lk(pl), known.astype(np.uint8)


def write_vcf(path, names, pl, known, true_geno, first_site=0):
    """The same sites as VCF text (CHROM 1, POS = 1 + site, REF A, ALT G, FORMAT GT:PL)."""
    gt = ["0/0", "0/1", "1/1"]
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.1\n")
        f.write('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n')
        f.write('##FORMAT=<ID=PL,Number=G,Type=Integer,Description="Phred-scaled genotype likelihoods">\n')
        f.write('##INFO=<ID=SYN,Number=0,Type=Flag,Description="synthetic">\n')
        f.write("##contig=<ID=1,length=249250621>\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
        for s in range(pl.shape[0]):
            site = first_site + s
            cols = ["1", str(1 + site), ("rs%d" % site) if known[s] else ".", "A", "G", "50", "PASS", "SYN", "GT:PL"]
            for i in range(pl.shape[1]):
                cols.append("%s:%d,%d,%d" % (gt[true_geno[s, i]], pl[s, i, 0], pl[s, i, 1], pl[s, i, 2]))
            f.write("\t".join(cols) + "\n")


# --------------------------------------------------------------------------------------
# The same generator with torch int64 arithmetic (wraps like uint64), so bench.py can
# create the 10M-site batch directly in HBM.  Bit-identical to gen_sites (tested).
# --------------------------------------------------------------------------------------
def _s64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def gen_batch_torch(mother, father, n_sites, config_no, first_site=0, device="cpu", chunk=1 << 20,
                    out_lk=None, out_flags=None):
    """-> (lk[float64 S,N,3], flags[uint8 S]) torch tensors on `device`."""
    import torch

    n = len(mother)
    D = 2 + 4 * n
    order = _topo(mother, father)
    seed = SEED_BASE + config_no
    g_, m1, m2 = _s64(int(GAMMA)), _s64(int(M1)), _s64(int(M2))
    lut = torch.from_numpy(PL_LUT).to(device)
    lk = out_lk if out_lk is not None else torch.empty((n_sites, n, 3), dtype=torch.float64, device=device)
    flags = out_flags if out_flags is not None else torch.empty(n_sites, dtype=torch.uint8, device=device)
    for lo in range(0, n_sites, chunk):
        hi = min(n_sites, lo + chunk)
        site = torch.arange(first_site + lo, first_site + hi, dtype=torch.int64, device=device)
        base = site * D

        def draw(k):
            z = (base + (k + 1)) * g_ + _s64(seed)
            z = (z ^ _lsr(z, 30)) * m1
            z = (z ^ _lsr(z, 27)) * m2
            return z ^ _lsr(z, 31)

        def u01(z):
            return _lsr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0)

        q = 0.01 + 0.49 * u01(draw(0))
        g = [None] * n
        for i in order:
            u1, u2 = u01(draw(1 + 2 * i)), u01(draw(2 + 2 * i))
            if mother[i] < 0:
                g[i] = (u1 < q).to(torch.int64) + (u2 < q).to(torch.int64)
            else:
                gm, gf = g[mother[i]], g[father[i]]
                am = torch.where(gm == 1, u1 < 0.5, gm == 2)
                af = torch.where(gf == 1, u2 < 0.5, gf == 2)
                g[i] = am.to(torch.int64) + af.to(torch.int64)
        flags[lo:hi] = (u01(draw(2 * n + 1)) < 0.10).to(torch.uint8)
        zero = torch.zeros(hi - lo, dtype=torch.int64, device=device)
        for i in range(n):
            r1 = _lsr(draw(2 * n + 2 + 2 * i), 33)
            r2 = _lsr(draw(2 * n + 3 + 2 * i), 33)
            a1 = 3 + r1 % 88
            a2 = 3 + r2 % 88
            far = torch.clamp(a1 + 10 + r2 % 156, max=255)
            t = g[i]
            p0 = torch.where(t == 0, zero, torch.where(t == 1, a1, far))
            p1 = torch.where(t == 1, zero, a1)
            p2 = torch.where(t == 2, zero, torch.where(t == 1, a2, far))
            lk[lo:hi, i, 0] = lut[p0]
            lk[lo:hi, i, 1] = lut[p1]
            lk[lo:hi, i, 2] = lut[p2]
    return lk, flags


# --------------------------------------------------------------------------------------
# Random pedigrees and adversarial likelihoods for the parity tests and the soak
# (tests/test_gpu_random_pedigrees.py, tests/test_gpu_soak.py, tools/soak_gpu.py); build()
# pre-compiles the generated kernels of the soak's pedigrees from the same seeds.
# --------------------------------------------------------------------------------------
def grow_pedigree(rng, n, allow_loops):
    """Start from founders, keep adding children of random (female, male) couples; without
    allow_loops a couple is only formed if it does not close a loop in the member/family graph."""
    ids, mids, fids, gen = [1, 2], [0, 0], [0, 0], [1, 2]
    comp = {1: 1, 2: 2}  # connected component of each member (for loop avoidance)
    couples = {}
    while len(ids) < n:
        r = rng.rand()
        females = [i for i, g in zip(ids, gen) if g == 2]
        males = [i for i, g in zip(ids, gen) if g == 1]
        if r < 0.3 or not females or not males:
            new = len(ids) + 1
            ids.append(new); mids.append(0); fids.append(0); gen.append(int(rng.randint(1, 3)))
            comp[new] = new
            continue
        mo, fa = int(rng.choice(females)), int(rng.choice(males))
        if (mo, fa) not in couples:
            if not allow_loops and comp[mo] == comp[fa]:
                continue
            couples[(mo, fa)] = True
            old = comp[fa]
            for k in comp:
                if comp[k] == old:
                    comp[k] = comp[mo]
        new = len(ids) + 1
        ids.append(new); mids.append(mo); fids.append(fa); gen.append(int(rng.randint(1, 3)))
        comp[new] = comp[mo]
    names = ["s%d" % i if rng.rand() < 0.75 else "NA" for i in ids]
    if all(x == "NA" for x in names):
        names[-1] = "s_last"
    return Pedigree(ids, mids, fids, gen, names)


def random_likelihoods(rng, ped, n_sites, max_pl=300):
    """Adversarial rows: PLs uniform in [0, max_pl) with no regard for Mendelian consistency, hard zeros, sites sharp enough
    for the -LRC shortcut, flags 0..3.  (Wide pedigrees want a smaller max_pl: fifty members that each contradict their
    parents at 1e-15 a piece put the whole site's probability mass below 1e-308, where every implementation's digits are
    what gradual underflow leaves of them.)"""
    pl = rng.randint(0, max_pl, size=(n_sites, ped.n, 3)).astype(float)
    pl[np.arange(n_sites)[:, None], np.arange(ped.n)[None, :], rng.randint(0, 3, size=(n_sites, ped.n))] = 0
    lk = 10.0 ** (-pl / 10.0)
    lk[rng.rand(n_sites, ped.n, 3) < 0.02] = 0.0           # hard zeros
    sharp = rng.rand(n_sites) < 0.15                        # some sites take the -LRC shortcut
    lk[sharp] = np.where(pl[sharp] == 0, 1.0, 1e-40)
    lk[:, ped.sequenced == 0, :] = 1.0
    return lk, rng.randint(0, 4, n_sites).astype(np.uint8)
