"""tools/gen_variant.py PED TAG [ENV=VALUE ...] — generate (plan-only, no GPU) the kernels of one
synthetic pedigree under tuning environment variables and drop sources + code objects into
tools/exp_variants/ as {el|ln}{N}_{TAG}_v{variant}.{hip,hsaco} for tools/exp_kb.sh."""
import glob, os, shutil, subprocess, sys, tempfile

ped_name, tag = sys.argv[1], sys.argv[2]
env = dict(os.environ)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    env[k] = v
tmp = tempfile.mkdtemp(prefix="famseq_gen_")
env.update(FAMSEQ_KERNEL_CACHE=tmp, FAMSEQ_KEEP_SRC="1", FAMSEQ_GEN_PED=ped_name)
code = r'''
import os, famseq_amd as fs
ped = fs.synthetic_pedigree(os.environ["FAMSEQ_GEN_PED"])
ctx = fs.Context(fs.make_model(ped), device=-1)
ctx.set_option("enum_impl", 1)
ctx.set_option("engine", fs.ENGINE_ELIM)
p = ctx.plan()
print("picked: elim variant", p["elim_variant"], p["elim_code_object"], "lane variant", p["enum_lane_variant"], p["enum_lane_code_object"])
'''
# the in-tree cache would win over FAMSEQ_KERNEL_CACHE for sources it already holds: fine, then nothing new to test
print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).stdout)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp_variants")
os.makedirs(out, exist_ok=True)
for src in glob.glob(tmp + "/*.hip"):
    head = open(src).readline()
    n = head.split(" for a ")[1].split("-member")[0]
    kind = "el" if "sum-product" in head else "ln"
    variant = head.strip().split("variant ")[-1].split(",")[0]
    base = "%s%s_%s_v%s" % (kind, n, tag, variant)
    shutil.copy(src, os.path.join(out, base + ".hip"))
    if os.path.exists(src[:-4] + ".hsaco"):
        shutil.copy(src[:-4] + ".hsaco", os.path.join(out, base + ".hsaco"))
    else:  # a variant that lost the pick: the cache keeps only its resource note — build it here to time it anyway
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--genco",
                               "-o", os.path.join(out, base + ".hsaco"), src])
    print(base, "scratch", open(src[:-4] + ".res").read().strip())
shutil.rmtree(tmp)
