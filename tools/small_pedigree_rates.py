import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, famseq_amd as fs
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream()
S = 8_000_000
for name in ('trio', 'quad', 'ped5'):
    ped = fs.synthetic_pedigree(name)
    n = ped.n
    mo, fa = ped.relations()
    lk, flags = fs.synth.gen_batch_torch(mo.tolist(), fa.tolist(), S, 1, device=dev)
    post, single = torch.empty_like(lk), torch.empty_like(lk)
    status = torch.empty(S, dtype=torch.uint8, device=dev)
    for eng, opt in (('enum', dict(enum_impl=1)), ('elim', dict(engine=fs.ENGINE_ELIM))):
        ctx = fs.Context(fs.make_model(ped), **opt)
        def step():
            ctx.bn_batch_device(S, lk.data_ptr(), flags.data_ptr(), post.data_ptr(), single.data_ptr(), status.data_ptr(), stream.cuda_stream)
        for _ in range(5): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(20): step()
        e1.record(stream); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        # stream probe on the same arrays
        nd = lk.numel()
        for _ in range(3): fs.stream_probe(ctx, nd, lk.data_ptr(), post.data_ptr(), single.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(10): fs.stream_probe(ctx, nd, lk.data_ptr(), post.data_ptr(), single.data_ptr(), stream.cuda_stream)
        e1.record(stream); torch.cuda.synchronize()
        pms = e0.elapsed_time(e1) / 10
        bps = 72 * n + 2
        print("%s %s: %.4f ms per %d sites, %.0f GB/s = %.3f of 8 TB/s; stream probe %.4f ms (%.0f GB/s): %.3f of probe" % (
            name, eng, ms, S, S * bps / ms / 1e6, S * bps / ms / 1e6 / 8000, pms, nd * 24 / pms / 1e6, (S * bps / ms) / (nd * 24 / pms)), flush=True)
        ctx.close()
