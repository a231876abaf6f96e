"""tools/host_path_trace.py — one famseq_bn_batch call from pinned host memory, to be run under
`rocprofv3 --kernel-trace --memory-copy-trace` so that the copy/kernel timeline of the chunked
two-stream pipeline can be read off the trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
import famseq_amd as fs

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
pin = (sys.argv[2] if len(sys.argv) > 2 else "pinned") == "pinned"
ped = fs.synthetic_pedigree("ped10")
mo, fa = ped.relations()
lk_t, fl_t = fs.synth.gen_batch_torch(mo.tolist(), fa.tolist(), S, 2, device="cuda")
mk = (lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory().numpy()) if pin else (lambda shape, dt: torch.empty(shape, dtype=dt).numpy())
lk = mk(tuple(lk_t.shape), torch.float64); lk[...] = lk_t.cpu().numpy()
fl = mk((S,), torch.uint8); fl[...] = fl_t.cpu().numpy()
post, single, st = mk(lk.shape, torch.float64), mk(lk.shape, torch.float64), mk((S,), torch.uint8)
ctx = fs.Context(fs.make_model(ped))
P = lambda x, t: x.ctypes.data_as(C.POINTER(t))
for _ in range(2):
    rc = fs.lib().famseq_bn_batch(ctx._h, S, P(lk, C.c_double), P(fl, C.c_uint8), P(post, C.c_double), P(single, C.c_double), P(st, C.c_uint8))
    assert rc == 0
print("done")
