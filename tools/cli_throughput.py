#!/usr/bin/env python3
"""End-to-end CLI throughput (timing point T3 of SURVEY.md 8(d)) on seeded synthetic ped10 sites.
  tools/cli_throughput.py [n_sites] [pedigree] VCF text through `FamSeq vcf`, `FamSeq pack`, then the packed paths
  tools/cli_throughput.py [n_sites] --packed   only the packed paths (`FamSeq PL`, `FamSeq PL -binOutput`); the packed
                                               input is written straight from the generator (no VCF text), so
                                               millions of sites take seconds to set up
Each command is run twice; the second (page cache and kernel cache warm) is reported, with the first in brackets."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from famseq_amd import synth, pedigree, plfile

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 500_000
packed_only = "--packed" in sys.argv
ped = pedigree.synthetic_pedigree(args[1] if len(args) > 1 else "ped10")  # tools/cli_throughput.py 5000000 trio
mo, fa = ped.relations()
d = tempfile.mkdtemp(prefix="fscli")
pedf, vcf, fspl = os.path.join(d, "p.ped"), os.path.join(d, "s.vcf"), os.path.join(d, "s.fspl")
pedigree.write_ped(ped, pedf)
cli = os.path.join(ROOT, "bin", "FamSeq")


def run(a, label, **env):
    ts = []
    for _ in range(2):
        out = a[a.index("-output") + 1]
        if os.path.exists(out):  # (overwriting a 1.5 GB file that sits in the page cache costs the second run 0.4 s of truncation:
            os.unlink(out)       # round 2's figures carried that; the file is removed outside the timed region now)
        t0 = time.time(); subprocess.check_call([cli] + a, stdout=subprocess.DEVNULL, env=dict(os.environ, FAMSEQ_TIMING="1", **env)); ts.append(time.time() - t0)
    print("%-14s %.2f s  %.2f M sites/s   [first run %.2f s]" % (label, ts[1], n / ts[1] / 1e6, ts[0]), flush=True)


if packed_only:
    t0 = time.time()
    pls, flags = [], []
    for lo in range(0, n, 1 << 20):
        pl, known, _ = synth.gen_sites(mo, fa, min(1 << 20, n - lo), synth.SEED_BASE + 2, lo)
        pls.append(pl.astype(np.uint16)); flags.append(known.astype(np.uint8))
    plfile.write_plfile(fspl, ped.names, np.concatenate(flags), np.concatenate(pls))
    print("wrote %d packed sites (%.0f MB) in %.1f s" % (n, os.path.getsize(fspl) / 1e6, time.time() - t0), flush=True)
else:
    pl, known, geno = synth.gen_sites(mo, fa, n, synth.SEED_BASE + 2)
    t0 = time.time(); synth.write_vcf(vcf, ped.names, pl, known, geno)
    print("wrote %d-site VCF (%.0f MB) in %.1f s" % (n, os.path.getsize(vcf) / 1e6, time.time() - t0), flush=True)
    run(["vcf", "-vcfFile", vcf, "-pedFile", pedf, "-output", os.path.join(d, "o.vcf")], "FamSeq vcf")
    run(["vcf", "-vcfFile", vcf, "-pedFile", pedf, "-output", os.path.join(d, "oh.vcf")], "vcf, host fmt", FAMSEQ_HOST_FORMAT="1")
    same = subprocess.call(["cmp", "-s", os.path.join(d, "o.vcf"), os.path.join(d, "oh.vcf")]) == 0
    print("device-formatted and host-formatted output identical:", same, flush=True)
    os.unlink(os.path.join(d, "oh.vcf"))
    run(["pack", "-vcfFile", vcf, "-pedFile", pedf, "-output", fspl], "FamSeq pack")
run(["PL", "-plFile", fspl, "-pedFile", pedf, "-output", os.path.join(d, "o.txt")], "FamSeq PL")
run(["PL", "-plFile", fspl, "-pedFile", pedf, "-output", os.path.join(d, "oh.txt")], "PL, host fmt", FAMSEQ_HOST_FORMAT="1")
os.unlink(os.path.join(d, "oh.txt"))
run(["PL", "-plFile", fspl, "-pedFile", pedf, "-output", os.path.join(d, "o.fspo"), "-binOutput"], "PL -binOutput")
run(["PL", "-plFile", fspl, "-pedFile", pedf, "-output", os.path.join(d, "o2.fspo"), "-binOutput", "-method", "2"], "PL -bin -m 2")
print("packed results: %.1f MB, packed input: %.1f MB" % (os.path.getsize(os.path.join(d, "o.fspo")) / 1e6, os.path.getsize(fspl) / 1e6))
