# tools/cli_threads.sh — `FamSeq vcf` on 3 M ten-member sites by parse / format thread counts (run on the GPU box)
set -u
python - <<'PY'
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from famseq_amd import synth, pedigree
n = 3_000_000
ped = pedigree.synthetic_pedigree("ped10")
mo, fa = ped.relations()
d = tempfile.mkdtemp(prefix="fscli")
pedf, vcf = os.path.join(d, "p.ped"), os.path.join(d, "s.vcf")
pedigree.write_ped(ped, pedf)
pl, known, geno = synth.gen_sites(mo, fa, n, synth.SEED_BASE + 2)
synth.write_vcf(vcf, ped.names, pl, known, geno)
out = os.path.join(d, "o.vcf")
for pt, ft in ((0, 0), (16, 16), (8, 16), (6, 16), (4, 16), (8, 12), (12, 16), (6, 12)):
    env = dict(os.environ, FAMSEQ_TIMING="1")
    if pt: env.update(FAMSEQ_PARSE_THREADS=str(pt), FAMSEQ_FORMAT_THREADS=str(ft))
    ts = []
    for _ in range(3):
        if os.path.exists(out): os.unlink(out)
        t0 = time.time(); r = subprocess.run(["bin/FamSeq", "vcf", "-vcfFile", vcf, "-pedFile", pedf, "-output", out], env=env, capture_output=True, text=True); ts.append(time.time() - t0)
    loop = [l for l in r.stderr.splitlines() if "loop" in l][-1]
    print("parse %2d format %2d: %.2f %.2f %.2f s  %s" % (pt, ft, ts[0], ts[1], ts[2], loop[:60] + " ... " + loop[loop.index("formatting"):loop.index("formatting") + 40]), flush=True)
PY
