import os, sys, glob, shutil, subprocess, tempfile
import famseq_amd as fs
from famseq_amd.prebuild_sets import wide_pedigree
name, tag = sys.argv[1], sys.argv[2]
from famseq_amd.prebuild_sets import soak_pedigree
ped = wide_pedigree(int(name[4:])) if name.startswith("wide") else (soak_pedigree(int(name[4:]))[1] if name.startswith("soak") else fs.synthetic_pedigree(name))
ctx = fs.Context(fs.make_model(ped), device=-1)
if ped.n <= 20: ctx.set_option("engine", fs.ENGINE_ELIM)
p = ctx.plan(); print(name, tag, "variant", p["elim_variant"], p["elim_code_object"])
obj = p["elim_code_object"]
out = "/root/repo/tools/exp_variants"
os.makedirs(out, exist_ok=True)
base = "el%d_%s_%s_v%d" % (ped.n, name, tag, p["elim_variant"])
shutil.copy(obj, out + "/" + base + ".hsaco"); shutil.copy(obj[:-6] + ".hip", out + "/" + base + ".hip")
print(base, "scratch", open(obj[:-6] + ".res").read().strip())
