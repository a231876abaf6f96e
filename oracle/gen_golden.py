#!/usr/bin/env python3
"""Generate tests/golden/* from the COMPILED REFERENCE (oracle/_ref).

Run in the build container (needs /root/reference):   python oracle/gen_golden.py
Everything written is data: inputs (likelihood arrays, copies of the reference's
TestData files) and the reference's outputs for them.  No reference source enters
the repo.  The fixtures are what pins oracle/bn_oracle.c and, through it, the HIP
path (tests/test_oracle_golden.py, tests/test_gpu_parity.py).

Fixtures
  golden/testdata/            fam01..06.ped, loftest.txt (verbatim TestData), and
                              test_subset.vcf = header + every PL-bearing line + every
                              97th other line of TestData/test.vcf (keeps CLI cases small)
  golden/ref_cli/             FamSeq_ref output for test_subset.vcf x fam01..06 (-method 1,
                              with -v and with -a) and loftest.txt x fam01..06 (LK mode)
  golden/bn_vcf.npz           per pedigree: lk/flags for the 12 computable sites of
                              test.vcf, reference post/single/status
  golden/bn_lk.npz            per pedigree: 100 loftest rows
  golden/bn_synth.npz         seeded ped5/ped10 batches, chrX, custom priors/mu/lc,
                              failure (status 1, status 2) and shortcut-boundary cases
  golden/tables.npz           transmission tables for several mutation rates
"""
import math
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from famseq_amd import pedigree, synth  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def pl_field_to_lk(s):
    # file.cpp:588-590: pow(10.0, -fabs(atof(field))/10.0)
    return math.pow(10.0, -abs(float(s)) / 10.0)


def vcf_sites(vcf_path, ped):
    """The computable sites of a VCF for one pedigree, by the reference's rules
    (file.cpp:362-593, 794-831): -> list of (pos, lk[N,3], flags)."""
    names = ped.names
    out = []
    with open(vcf_path) as f:
        for line in f:
            line = line.rstrip("\n")
            if len(line) < 2:
                break
            if line.startswith("##"):
                continue
            t = line.split("\t")
            if line.startswith("#CHROM"):
                cols = t[9:]
                v2p = [names.index(c) if c in names else -1 for c in cols]
                continue
            if t[3] in (".", "-") or len(t[3]) != 1 or len(t[4]) != 1:
                continue
            chrom = t[0]
            if chrom in ("Y", "chrY", "MT"):
                continue
            c = chrom[3:] if chrom.startswith("chr") else chrom
            try:
                cn = int(c)
            except ValueError:
                cn = 0
            is_x = chrom in ("X", "chrX", "CHRX")
            if not (0 < cn < 23 or is_x):
                continue
            fmt = t[8].split(":")
            seq_cols = [i for i, p in enumerate(v2p) if p >= 0]
            miss = sum(1 for i in seq_cols if len(t[9 + i]) < 5)
            if miss == len(seq_cols):
                continue
            ipl = -1
            for k, key in enumerate(fmt):
                if key in ("PL", "GL"):
                    ipl = k
            if ipl < 0:
                continue
            lk = np.ones((ped.n, 3))
            for i in seq_cols:
                if len(t[9 + i]) < 5:
                    continue
                sub = t[9 + i].split(":")
                if len(sub) != len(fmt):
                    continue
                pls = sub[ipl].split(",")
                for g in range(3):
                    lk[v2p[i], g] = pl_field_to_lk(pls[g])
            flags = (1 if t[2] != "." else 0) | (2 if is_x else 0)
            out.append((int(t[1]), lk, flags))
    return out


def lk_rows(path, ped):
    """LK file rows (file.cpp:1640-1751, -lkType n): -> lk[R,N,3]."""
    rows = []
    with open(path) as f:
        cols = f.readline().rstrip("\n").split("\t")
        v2p = [ped.names.index(c) if c in ped.names else -1 for c in cols]
        for line in f:
            line = line.rstrip("\n")
            if len(line) < 2:
                break
            t = line.split("\t")
            lk = np.ones((ped.n, 3))
            for i, p in enumerate(v2p):
                if p >= 0:
                    lk[p] = [float(x) for x in t[i].split(",")]
            rows.append(lk)
    return np.array(rows)


def ref_family(ped, **kw):
    return oracle.RefFamily(ped.ids, ped.mids, ped.fids, ped.genders, ped.sequenced, **kw)


def pack(store, key, ped, lk, flags, post, single, status, **extra):
    store[key + ".lk"] = lk
    store[key + ".flags"] = np.asarray(flags, np.uint8)
    store[key + ".post"] = post
    store[key + ".single"] = single
    store[key + ".status"] = status
    store[key + ".ped"] = np.array([ped.ids, ped.mids, ped.fids, ped.genders], np.int32)
    store[key + ".sequenced"] = ped.sequenced
    for k, v in extra.items():
        store[key + "." + k] = np.asarray(v)


def main():
    assert oracle.have_ref(), "run `make -C oracle ref` first (needs /root/reference)"
    os.makedirs(os.path.join(OUT, "testdata"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "ref_cli"), exist_ok=True)

    # ---- TestData copies + reduced VCF ---------------------------------
    for k in range(1, 7):
        shutil.copyfile("%s/TestData/fam%02d.ped" % (REF, k), "%s/testdata/fam%02d.ped" % (OUT, k))
    shutil.copyfile(REF + "/TestData/loftest.txt", OUT + "/testdata/loftest.txt")
    sub = os.path.join(OUT, "testdata", "test_subset.vcf")
    with open(REF + "/TestData/test.vcf") as f, open(sub, "w") as g:
        other = 0
        for line in f:
            if line.startswith("#"):
                g.write(line)
                continue
            fmt = line.split("\t")[8]
            if "PL" in fmt.split(":"):
                g.write(line)
            else:
                other += 1
                if other % 97 == 0:
                    g.write(line)

    # ---- reference CLI text outputs ------------------------------------
    for k in range(1, 7):
        pedf = "%s/testdata/fam%02d.ped" % (OUT, k)
        for tag, extra in (("v", ["-v"]), ("a", ["-a"]), ("plain", [])):
            outp = "%s/ref_cli/subset_fam%02d_%s.vcf" % (OUT, k, tag)
            subprocess.check_call([oracle.REF_CLI, "vcf", "-vcfFile", sub, "-pedFile", pedf, "-output", outp,
                                   "-method", "1"] + extra, stdout=subprocess.DEVNULL)
        outp = "%s/ref_cli/loftest_fam%02d.txt" % (OUT, k)
        subprocess.check_call([oracle.REF_CLI, "LK", "-lkFile", OUT + "/testdata/loftest.txt", "-pedFile", pedf,
                               "-output", outp, "-method", "1"], stdout=subprocess.DEVNULL)
    # full test.vcf result lines for fam01 (-v) = BASELINE config #1 text pin
    full = OUT + "/ref_cli/full_fam01_v.vcf"
    subprocess.check_call([oracle.REF_CLI, "vcf", "-vcfFile", REF + "/TestData/test.vcf", "-pedFile",
                           OUT + "/testdata/fam01.ped", "-output", full, "-method", "1", "-v"], stdout=subprocess.DEVNULL)
    # BASELINE config #1 at full length: the whole TestData/test.vcf (a data file of the reference's
    # own tests, 2 MB, kept gzipped) and the reference CLI's plain (9,749 lines) and -a (10,008 lines)
    # outputs for fam01 — pins the header echo / one-line-lag logic and the echo / drop rules
    # (file.cpp:143-196, :362-555) over every line, not 1 in 97 of them
    import gzip

    def gz(src, dst):
        with open(src, "rb") as f, open(dst, "wb") as raw, gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as g:
            shutil.copyfileobj(f, g)

    gz(REF + "/TestData/test.vcf", OUT + "/testdata/test_full.vcf.gz")
    for tag, extra in (("plain", []), ("a", ["-a"])):
        tmp = "/tmp/famseq_full_%s.vcf" % tag
        subprocess.check_call([oracle.REF_CLI, "vcf", "-vcfFile", REF + "/TestData/test.vcf", "-pedFile",
                               OUT + "/testdata/fam01.ped", "-output", tmp, "-method", "1"] + extra, stdout=subprocess.DEVNULL)
        gz(tmp, "%s/ref_cli/full_fam01_%s.vcf.gz" % (OUT, tag))
        os.unlink(tmp)

    # ---- probe inputs for the CLI surface: chrX, Known, failures,
    # missing sample, GL, multi-base REF, chrY/MT/odd contigs, location file, custom flags
    probe_ped = OUT + "/testdata/probe.ped"
    with open(probe_ped, "w") as f:
        f.write("ID\tmID\tfID\tgender IndividualName\n1\t0\t0\t1\tpa\n2\t0\t0\t2\tma\n3\t2\t1\t1\tson\n4\t2\t1\t2\tdau")
    rows = [
        ("1", 100, ".", "A", "G", "GT:PL", ["0/1:30,0,40", "0/0:0,25,200", "0/1:20,0,35", "0/0:0,12,90"]),
        ("X", 200, ".", "A", "G", "GT:PL", ["1/1:90,30,0", "0/1:25,0,30", "0/0:0,20,80", "0/1:15,0,40"]),
        ("chrX", 300, "rs1", "A", "G", "GT:PL", ["0/1:50,0,50", "0/1:25,0,30", "0/0:0,20,80", "0/1:15,0,40"]),
        ("X", 400, ".", "A", "G", "GT:PL", ["0/1:9999,0,9999", "0/1:25,0,30", "0/0:0,20,80", "0/1:15,0,40"]),
        ("1", 500, ".", "A", "G", "GT:PL", ["1/1:9999,9999,0", "1/1:9999,9999,0", "0/0:0,9999,9999", "0/0:0,12,90"]),
        ("1", 600, ".", "A", "G", "GT:PL", ["./.", "0/0:0,30,60", "0/0:0,25,50", "0/0:0,12,90"]),
        ("1", 700, ".", "A", "G", "GT:GL", ["0/1:-3.0,0,-4.0", "0/0:0,-2.5,-20.0", "0/1:-2.0,0,-3.5", "0/0:0,-1.2,-9.0"]),
        ("1", 800, "rs8", "AT", "G", "GT:PL", ["0/1:30,0,40", "0/0:0,25,200", "0/1:20,0,35", "0/0:0,12,90"]),
        ("1", 900, ".", "A", ".", "GT:PL", ["0/0:0,30,400", "0/0:0,25,200", "0/0:0,20,350", "0/0:0,12,90"]),
        ("Y", 1000, ".", "A", "G", "GT:PL", ["0/1:30,0,40", "0/0:0,25,200", "0/1:20,0,35", "0/0:0,12,90"]),
        ("MT", 1100, ".", "A", "G", "GT:PL", ["0/1:30,0,40", "0/0:0,25,200", "0/1:20,0,35", "0/0:0,12,90"]),
        ("GL000207.1", 1200, ".", "A", "G", "GT:PL", ["0/1:30,0,40", "0/0:0,25,200", "0/1:20,0,35", "0/0:0,12,90"]),
        ("chr7", 1300, "rs13", "C", "T", "GT:DP", ["0/1:30", "0/0:25", "0/1:20", "0/0:12"]),
        ("22", 1400, ".", "A", "G", "GT:PL", ["./.", "./.", "./.", "./."]),
        ("2", 1500, ".", "A", "G", "GT:AD:PL", ["0/1:3,4:300,0,400", "0/0:0,250,2000", "0/1:1,1:200,0,350", "0/0:9,0:0,120,900"]),
        ("3", 1600, "rs16", "G", "C", "GT:PL", ["0/0:0,200,2000", "0/0:0,180,1800", "0/0:0,170,1700", "0/0:0,165,1650"]),
    ]
    probe = OUT + "/testdata/probe.vcf"
    with open(probe, "w") as f:
        f.write("##fileformat=VCFv4.1\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n")
        f.write("##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"PL\">\n##INFO=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n")
        f.write("##contig=<ID=1,length=249250621>\n##contig=<ID=X,length=155270560>\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tdau\tother\tpa\tson\tma\n")
        for c, pos, rid, ref, alt, fmt, smp in rows:
            pa, ma, son, dau = smp
            f.write("\t".join([c, str(pos), rid, ref, alt, "50", "PASS", "DP=9", fmt, dau, "0/0:0,9,99", pa, son, ma]) + "\n")
    noanchor = OUT + "/testdata/probe_noanchor.vcf"  # no ##INFO / ##contig anchors: fall-back header lines
    with open(noanchor, "w") as f:
        f.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tpa\tma\tson\tdau\n")
        for c, pos, rid, ref, alt, fmt, smp in rows[:3]:
            f.write("\t".join([c, str(pos), rid, ref, alt, "50", "PASS", "DP=9", fmt] + smp) + "\n")
    with open(OUT + "/testdata/probe.loc", "w") as f:
        f.write("1\t100\nX\t300\n1\t600\n2\t1500\n5\t77\n")
    probe_runs = {
        "default": [], "a": ["-a"], "v": ["-v"], "mu0": ["-mRate", "0", "-a"],
        "priors": ["-genoProbN", "0.9", "0.08", "0.02", "-genoProbK", "0.3", "0.4", "0.3", "-genoProbXN", "0.97", "0.03",
                   "-genoProbXK", "0.6", "0.4", "-mRate", "1e-3", "-LRC", "0.9"],
        "loc": ["-l", OUT + "/testdata/probe.loc"],
    }
    for tag, extra in probe_runs.items():
        subprocess.check_call([oracle.REF_CLI, "vcf", "-vcfFile", probe, "-pedFile", probe_ped, "-output",
                               "%s/ref_cli/probe_%s.vcf" % (OUT, tag), "-method", "1"] + extra, stdout=subprocess.DEVNULL)
    subprocess.check_call([oracle.REF_CLI, "vcf", "-vcfFile", noanchor, "-pedFile", probe_ped, "-output",
                           OUT + "/ref_cli/probe_noanchor.vcf"], stdout=subprocess.DEVNULL)
    # -method 2 (Elston-Stewart) outputs: our CLI serves that flag with the exact sum-product engine
    for tag, vcf_in, pedf in (("probe", probe, probe_ped), ("subset_fam01", sub, OUT + "/testdata/fam01.ped"),
                              ("subset_fam04", sub, OUT + "/testdata/fam04.ped")):
        subprocess.check_call([oracle.REF_CLI, "vcf", "-vcfFile", vcf_in, "-pedFile", pedf, "-output",
                               "%s/ref_cli/%s_method2.vcf" % (OUT, tag), "-method", "2", "-a"], stdout=subprocess.DEVNULL)
    subprocess.check_call([oracle.REF_CLI, "LK", "-lkFile", OUT + "/testdata/loftest.txt", "-pedFile", OUT + "/testdata/fam01.ped",
                           "-output", OUT + "/ref_cli/loftest_fam01_method2.txt", "-method", "2"], stdout=subprocess.DEVNULL)
    # LK files in the three transformed likelihood types, from the first 20 loftest rows
    with open(OUT + "/testdata/loftest.txt") as f:
        lk_lines = [l.rstrip("\n") for l in f][:21]
    conv = {"log10": lambda x: "%.10g" % math.log10(x), "ln": lambda x: "%.10g" % math.log(x),
            "PS": lambda x: "%.10g" % (-10 * math.log10(x))}
    for typ, fn in conv.items():
        path = "%s/testdata/lk_%s.txt" % (OUT, typ)
        with open(path, "w") as f:
            f.write(lk_lines[0] + "\n")
            for l in lk_lines[1:]:
                f.write("\t".join(",".join(fn(float(x)) for x in cell.split(",")) if cell else "" for cell in l.split("\t")) + "\n")
        subprocess.check_call([oracle.REF_CLI, "LK", "-lkFile", path, "-pedFile", OUT + "/testdata/fam04.ped", "-output",
                               "%s/ref_cli/lk_%s_fam04.txt" % (OUT, typ), "-lkType", typ], stdout=subprocess.DEVNULL)

    # ---- raw fp64 pins ---------------------------------------------------
    vcf, lkf = {}, {}
    for k in range(1, 7):
        ped = pedigree.read_ped("%s/testdata/fam%02d.ped" % (OUT, k))
        fam = ref_family(ped)
        sites = vcf_sites(REF + "/TestData/test.vcf", ped)
        lk = np.array([s[1] for s in sites])
        flags = np.array([s[2] for s in sites], np.uint8)
        post, single, st = fam.bn_batch(lk, flags)
        p2, _, st2 = fam.bn_batch(lk, flags, method=2)
        pack(vcf, "fam%02d" % k, ped, lk, flags, post, single, st, pos=[s[0] for s in sites], peel=p2)
        rows = lk_rows(OUT + "/testdata/loftest.txt", ped)
        post, single, st = fam.bn_batch(rows)
        p2, _, _ = fam.bn_batch(rows, method=2)
        pack(lkf, "fam%02d" % k, ped, rows, np.zeros(len(rows), np.uint8), post, single, st, peel=p2)
        print("fam%02d: N=%d  vcf sites=%d (full BN %d)  lk rows=%d" % (
            k, ped.n, len(sites), int(np.sum((vcf["fam%02d.status" % k] & 0x80) == 0)), len(rows)))
    np.savez_compressed(OUT + "/bn_vcf.npz", **vcf)
    np.savez_compressed(OUT + "/bn_lk.npz", **lkf)

    # ---- synthetic -------------------------------------------------------
    syn = {}
    rng = np.random.RandomState(20240501)
    for name, cfg, ns in (("ped5", 1, 256), ("ped10", 2, 48)):
        ped = pedigree.synthetic_pedigree(name)
        mo, fa = ped.relations()
        lk, flags = synth.gen_batch(mo, fa, ns, cfg)
        post, single, st = ref_family(ped).bn_batch(lk, flags)
        pack(syn, name, ped, lk, flags, post, single, st)
    # chrX on ped10 and ped5, mixed Known, same PL model
    for name, cfg, ns in (("ped5", 11, 128), ("ped10", 12, 24)):
        ped = pedigree.synthetic_pedigree(name)
        mo, fa = ped.relations()
        lk, flags = synth.gen_batch(mo, fa, ns, cfg)
        flags = flags | 2
        post, single, st = ref_family(ped).bn_batch(lk, flags)
        pack(syn, name + "_x", ped, lk, flags, post, single, st)
    # custom constants: mu, lc, priors
    ped = pedigree.synthetic_pedigree("ped5")
    mo, fa = ped.relations()
    lk, flags = synth.gen_batch(mo, fa, 96, 21)
    flags = flags | (rng.randint(0, 2, 96).astype(np.uint8) << 1)
    kw = dict(mrate=1e-3, lc=0.9, genoProbN=[0.9, 0.08, 0.02], genoProbK=[0.3, 0.4, 0.3],
              genoProbXN=[0.97, 0, 0.03], genoProbXK=[0.6, 0, 0.4])
    post, single, st = ref_family(ped, **kw).bn_batch(lk, flags)
    pack(syn, "ped5_custom", ped, lk, flags, post, single, st, mrate=kw["mrate"], lc=kw["lc"],
         gN=kw["genoProbN"], gK=kw["genoProbK"], gXN=kw["genoProbXN"], gXK=kw["genoProbXK"])
    # probe family of SURVEY App. H: pa ma son dau; extremes, failures, missing
    quad = pedigree.Pedigree([1, 2, 3, 4], [0, 0, 2, 2], [0, 0, 1, 1], [1, 2, 1, 2], ["pa", "ma", "son", "dau"])

    def L(*pls):
        return np.array([[pl_field_to_lk(x) for x in row] for row in pls])

    probes = [
        (L((30, 0, 40), (0, 25, 200), (20, 0, 35), (0, 12, 90)), 0),
        (L((90, 30, 0), (25, 0, 30), (0, 20, 80), (15, 0, 40)), 2),
        (L((50, 0, 50), (25, 0, 30), (0, 20, 80), (15, 0, 40)), 3),
        (L((9999, 0, 9999), (25, 0, 30), (0, 20, 80), (15, 0, 40)), 2),  # status 1 on chrX
        (L((9999, 9999, 0), (9999, 9999, 0), (0, 9999, 9999), (0, 12, 90)), 0),
        (np.vstack([np.ones((1, 3)), L((0, 30, 60), (0, 25, 50), (0, 12, 90))]), 0),  # pa missing
        (L((0, 3233, 3233), (0, 3234, 5000), (0, 400, 400), (0, 160, 160)), 0),  # shortcut
        (L((0, 159, 400), (0, 400, 400), (0, 400, 400), (0, 400, 400)), 1),  # just below the shortcut
        (L((0, 160, 160), (0, 160, 160), (0, 160, 160), (0, 160, 160)), 1),  # shortcut boundary
        (L((0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0)), 0),  # flat
        (np.zeros((4, 3)), 0),  # all-zero likelihood -> status 1
        (L((0, 84692, 84692), (84692, 0, 84692), (0, 57, 1457), (808, 0, 634)), 1),  # test.vcf-scale PLs
    ]
    lk = np.array([p[0] for p in probes])
    flags = np.array([p[1] for p in probes], np.uint8)
    post, single, st = ref_family(quad).bn_batch(lk, flags)
    pack(syn, "quad", quad, lk, flags, post, single, st)
    post, single, st = ref_family(quad, mrate=0.0).bn_batch(lk, flags)  # status 2: hard zeros
    pack(syn, "quad_mu0", quad, lk, flags, post, single, st, mrate=0.0)
    # shortcut boundary sweep on a trio: PL 150..165 for both non-zero genotypes
    trio = pedigree.Pedigree([1, 2, 3], [0, 0, 2], [0, 0, 1], [1, 2, 2], ["f", "m", "c"])
    lk = np.array([L((0, a, a), (0, b, b + 7), (0, a + 1, 300)) for a in range(150, 166) for b in (155, 160, 3000)])
    flags = np.arange(len(lk), dtype=np.uint8) % 2
    post, single, st = ref_family(trio).bn_batch(lk, flags)
    pack(syn, "trio_lrc", trio, lk, flags, post, single, st)
    # partially sequenced + chain pedigree (few leaves) + unrelated founders
    chain = pedigree.Pedigree([1, 2, 3, 4, 5, 6, 7], [0, 0, 2, 0, 4, 0, 6], [0, 0, 1, 0, 3, 0, 5],
                              [1, 2, 1, 2, 1, 2, 2], ["NA", "a", "NA", "b", "c", "NA", "d"])
    mo, fa = chain.relations()
    lk, flags = synth.gen_batch(mo, fa, 64, 31)
    lk[:, chain.sequenced == 0, :] = 1.0
    flags = flags | ((np.arange(64) % 3 == 0).astype(np.uint8) << 1)
    post, single, st = ref_family(chain).bn_batch(lk, flags)
    pack(syn, "chain7", chain, lk, flags, post, single, st)
    solo = pedigree.Pedigree([1, 2, 3], [0, 0, 0], [0, 0, 0], [1, 2, 1], ["a", "b", "c"])
    lk, flags = synth.gen_batch(*solo.relations(), 32, 32)
    post, single, st = ref_family(solo).bn_batch(lk, flags)
    pack(syn, "founders3", solo, lk, flags, post, single, st)
    np.savez_compressed(OUT + "/bn_synth.npz", **syn)

    # ---- tables -----------------------------------------------------------
    tab = {}
    for mu in (1e-7, 0.0, 1e-3, 0.01, 0.5, 2.5e-8):
        fam = oracle.RefFamily([1, 2, 3], [0, 0, 2], [0, 0, 1], [1, 2, 1], mrate=mu)
        a, b, c = fam.tables()
        tab["mu%g.pcp2" % mu], tab["mu%g.xf" % mu], tab["mu%g.xm" % mu] = a, b, c
    np.savez_compressed(OUT + "/tables.npz", **tab)
    for k in sorted(syn):
        if k.endswith(".status"):
            print(k, np.unique(syn[k], return_counts=True))
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
