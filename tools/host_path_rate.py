#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point famseq_bn_batch:
the same seeded ped10 batch from pageable and from pinned host memory."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import famseq_amd as fs

name, S = (sys.argv[1] if len(sys.argv) > 1 else "ped10"), int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
ped = fs.synthetic_pedigree(name)
mo, fa = ped.relations()
cfg = {"ped5": 1, "ped10": 2, "ped15": 4}[name]
lk_t, fl_t = fs.synth.gen_batch_torch(mo.tolist(), fa.tolist(), S, cfg, device="cuda")
lk, fl = lk_t.cpu().numpy(), fl_t.cpu().numpy()
ctx = fs.Context(fs.make_model(ped))
# what the link itself gives (pinned, one direction at a time, then both at once on two streams)
h = torch.empty(1 << 27, dtype=torch.uint8).pin_memory()
d = torch.empty(1 << 27, dtype=torch.uint8, device="cuda")
h2, d2 = torch.empty_like(h).pin_memory(), torch.empty_like(d)
def bw(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return n * (1 << 27) / (time.perf_counter() - t0) / 1e9
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
print("link: H2D %.1f GB/s, D2H %.1f GB/s, both at once %.1f GB/s each" % (bw(lambda: d.copy_(h, non_blocking=True)), bw(lambda: h.copy_(d, non_blocking=True)), bw(both)))
del h, d, h2, d2
chunks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
for label, pin, chunk in [(l, p_, c) for c in chunks for (l, p_) in (("pageable", False), ("pinned", True))]:
    ctx.set_option("chunk_sites", chunk)
    label = "%s chunk=%s" % (label, chunk or "default")
    if pin:
        bufs = [torch.empty(lk.shape, dtype=torch.float64).pin_memory() for _ in range(3)]
        bufs[0].copy_(torch.from_numpy(lk))
        a = [b.numpy() for b in bufs]
        st = torch.empty(S, dtype=torch.uint8).pin_memory().numpy()
        flp = torch.from_numpy(fl).pin_memory().numpy()
    else:
        a = [lk, np.empty_like(lk), np.empty_like(lk)]
        st, flp = np.empty(S, np.uint8), fl
    import ctypes as C
    P = lambda x, t: x.ctypes.data_as(C.POINTER(t))
    def call():
        rc = fs.lib().famseq_bn_batch(ctx._h, S, P(a[0], C.c_double), P(flp, C.c_uint8), P(a[1], C.c_double), P(a[2], C.c_double), P(st, C.c_uint8))
        assert rc == 0
    call()
    t0 = time.perf_counter(); call(); call(); dt = (time.perf_counter() - t0) / 2
    print("%s %s: %.1f M sites/s, %.2f GB/s over the host link (%d B/site)" % (name, label, S / dt / 1e6, S * (72 * ped.n + 2) / dt / 1e9, 72 * ped.n + 2))
