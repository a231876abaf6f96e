"""tools/call_rate_resident.py — the call-path kernels of the ten-member pedigree (enumeration and sum-product forms) on resident buffers,
twenty launches back to back (famseq_bn_call_batch_device, HIP events): ms per 1 M sites.  (Under rocprofv3 with a transfer
between launches — tools/call_now.sh — the same kernels read 5-20 % slower: the clocks of a GPU that idles between launches.)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np, famseq_amd as fs
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream()
ped = fs.synthetic_pedigree("ped10"); n = ped.n; mo, fa = ped.relations()
for S in (1_000_000, 4_000_000):
    pl, known, _ = fs.synth.gen_sites(mo, fa, S, fs.synth.SEED_BASE + 2)
    d_pl = torch.from_numpy(pl.astype(np.uint16).view(np.int16)).to(dev); d_fl = torch.from_numpy(known.astype(np.uint8)).to(dev)
    d_gpp = torch.empty((S, n, 3), dtype=torch.float64, device=dev); d_fpp = torch.empty_like(d_gpp)
    d_fgt = torch.empty((S, n), dtype=torch.int8, device=dev); d_st = torch.empty(S, dtype=torch.uint8, device=dev)
    seq = np.arange(n, dtype=np.int32)
    for eng, opt in (("enum", dict(enum_impl=1)), ("elim", dict(engine=fs.ENGINE_ELIM))):
        ctx = fs.Context(fs.make_model(ped), **opt)
        def step():
            ctx.bn_call_batch_device(S, seq, d_pl16=d_pl.data_ptr(), d_flags=d_fl.data_ptr(), d_gpp=d_gpp.data_ptr(), d_fpp=d_fpp.data_ptr(), d_fgt=d_fgt.data_ptr(), d_status=d_st.data_ptr(), stream=stream.cuda_stream)
        for _ in range(5): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(20): step()
        e1.record(stream); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("ped10 call path %s: %.4f ms per %d sites = %.4f ms per 1 M" % (eng, ms, S, ms * 1e6 / S), flush=True)
        ctx.close()
