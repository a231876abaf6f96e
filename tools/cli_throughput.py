#!/usr/bin/env python3
"""End-to-end CLI throughput (timing point T3 of SURVEY.md 8(d)): seeded synthetic ped10 VCF text
through `FamSeq vcf`, the same sites packed through `FamSeq PL`, and (reference point) `FamSeq pack`."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from famseq_amd import synth, pedigree

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
ped = pedigree.synthetic_pedigree("ped10")
mo, fa = ped.relations()
d = tempfile.mkdtemp(prefix="fscli")
pedf, vcf = os.path.join(d, "p.ped"), os.path.join(d, "s.vcf")
pedigree.write_ped(ped, pedf)
pl, known, geno = synth.gen_sites(mo, fa, n, synth.SEED_BASE + 2)
t0 = time.time(); synth.write_vcf(vcf, ped.names, pl, known, geno); print("wrote %d-site VCF (%.0f MB) in %.1f s" % (n, os.path.getsize(vcf) / 1e6, time.time() - t0))
cli = os.path.join(ROOT, "bin", "FamSeq")
def run(args, label):
    t0 = time.time(); subprocess.check_call([cli] + args, stdout=subprocess.DEVNULL); dt = time.time() - t0
    print("%-12s %.2f s  %.2f M sites/s" % (label, dt, n / dt / 1e6))
run(["vcf", "-vcfFile", vcf, "-pedFile", pedf, "-output", os.path.join(d, "o.vcf")], "FamSeq vcf")
run(["pack", "-vcfFile", vcf, "-pedFile", pedf, "-output", os.path.join(d, "s.fspl")], "FamSeq pack")
run(["PL", "-plFile", os.path.join(d, "s.fspl"), "-pedFile", pedf, "-output", os.path.join(d, "o.txt")], "FamSeq PL")
run(["PL", "-plFile", os.path.join(d, "s.fspl"), "-pedFile", pedf, "-output", os.path.join(d, "o.fspo"), "-binOutput"], "PL -binOutput")
print("packed results: %.1f MB" % (os.path.getsize(os.path.join(d, "o.fspo")) / 1e6))
print("packed file: %.1f MB" % (os.path.getsize(os.path.join(d, "s.fspl")) / 1e6))
