"""tools/phase_clock_plain.py [pedigree ...] — FAMSEQ_PHASE_CLOCK=1 on the plain generated kernels (enumeration, sum-product): shares of
the waves' cycles per phase of the chunk loop (0 stage in [+ rows to registers], 1 single posterior, 2 single rows out, 3 engine body,
5 marginals out [r = 0 shells: 6 vote, 3 body, 1 single, 2 out, 4 rows, 5 out]).  Run on the GPU box with its own kernel cache.
INDICATIVE ONLY for these kernels: every mark is an s_memtime behind an s_waitcnt lgkmcnt(0), which costs the short phases of a
memory-bound kernel their overlap — the marked kernels run 2-3 x slower than the plain ones (the call-path kernel: 1.4 x).  Seen
(quad / five members, enumeration): stage-in 20 / 38 %, single posterior 27 / 22 %, body 28 / 19 %, the two stage-outs 25 / 21 %."""
import os, sys
os.environ["FAMSEQ_PHASE_CLOCK"] = "1"
os.environ.setdefault("FAMSEQ_KERNEL_CACHE", "/tmp/kc_phase_plain")
os.makedirs(os.environ["FAMSEQ_KERNEL_CACHE"], exist_ok=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, famseq_amd as fs
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
for name in (sys.argv[1:] or ["quad", "ped5", "ped10"]):
    ped = fs.synthetic_pedigree(name)
    S = 4_000_000 if ped.n <= 6 else 1_000_000
    mo, fa = ped.relations()
    lk, flags = fs.synth.gen_batch_torch(mo.tolist(), fa.tolist(), S, 1, device=dev)
    post, single = torch.empty_like(lk), torch.empty_like(lk)
    status = torch.empty(S, dtype=torch.uint8, device=dev)
    for eng, opt in (("enum", dict(enum_impl=1)), ("elim", dict(engine=fs.ENGINE_ELIM))):
        if eng == "enum" and ped.n > 12:
            continue
        ctx = fs.Context(fs.make_model(ped), **opt)
        for _ in range(3):
            ctx.bn_batch_device(S, lk.data_ptr(), flags.data_ptr(), post.data_ptr(), single.data_ptr(), status.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        ctx.set_option("phase_clock_report", 1)  # discard the warm-up's counts
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            ctx.bn_batch_device(S, lk.data_ptr(), flags.data_ptr(), post.data_ptr(), single.data_ptr(), status.data_ptr(), stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        print("%s %s: %.4f ms per %d sites (with the marks)" % (name, eng, e0.elapsed_time(e1) / 5, S), file=sys.stderr, flush=True)
        ctx.set_option("phase_clock_report", 1)
        ctx.close()
