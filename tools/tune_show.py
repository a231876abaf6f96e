import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ["FAMSEQ_KERNEL_CACHE"] = "/tmp/kc_tune2"
import famseq_amd as fs
for name in ("ped15:12", "ped15", "ped10", "ped15:9", "sib8", "ped5"):
    ctx = fs.Context(fs.make_model(fs.synthetic_pedigree(name)))
    ctx.set_option("tune", 1)
    print(name, "|", ctx.plan()["tune"])
    ctx.close()
