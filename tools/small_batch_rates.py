"""tools/small_batch_rates.py [workload] — enumeration time per batch for small batches (resident
inputs, HIP events; median of 20): the compiled-in team-per-site kernel, the generated kernel with one
lane per site, each lanes-per-site group size, and what auto mode picks.  Run on the GPU box."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import famseq_amd as fs  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ped10"
ped = fs.synthetic_pedigree(name)
mo, fa = ped.relations()
dev = torch.device("cuda", 0)
sizes = [1, 16, 64, 256, 1000, 4000, 14000, 32000, 65536, 131072]
lk, flags = fs.synth.gen_batch_torch(mo.tolist(), fa.tolist(), max(sizes), 2, device=dev)
post, single = torch.empty_like(lk), torch.empty_like(lk)
status = torch.empty(max(sizes), dtype=torch.uint8, device=dev)
model = fs.make_model(ped)
probe = fs.Context(model, device=-1)
dmax = probe.plan()["enum_group_digits_max"]
probe.close()
modes = [("team", dict(enum_impl=0)), ("lane d=0", dict(enum_impl=1, group_digits=0))]
modes += [("group d=%d" % d, dict(enum_impl=1, group_digits=d)) for d in range(1, dmax + 1)]
modes += [("auto", dict())]  # the defaults: a generated kernel for any batch size once its code object is loaded or on disk
stream = torch.cuda.current_stream()
out = {}
for label, opt in modes:
    ctx = fs.Context(model, **opt)
    row = {}
    for n in sizes:
        def step():
            ctx.bn_batch_device(n, lk.data_ptr(), flags.data_ptr(), post.data_ptr(), single.data_ptr(), status.data_ptr(),
                                stream.cuda_stream)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); step(); b.record(stream)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        row[n] = float(np.median(ts))
        if label == "auto":
            row["d@%d" % n] = ctx.plan()["enum_group_digits_last"]
    out[label] = row
    print("%-10s" % label, "  ".join("%s:%.4f" % (k, v) if isinstance(v, float) else "%s:%s" % (k, v) for k, v in row.items()), flush=True)
    ctx.close()
json.dump(out, open("gpurun_out/small_batch_rates_%s.json" % name, "w"), indent=1)
