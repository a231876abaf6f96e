"""famseq_amd — MI355X-native FamSeq `-method 1` pedigree posterior (host-side Python binding).

The product is ``famseq_amd/lib/libfamseq_hip.so`` (HIP, gfx950; C ABI in
``include/famseq_hip.h``).  This module is a thin ctypes layer over that ABI whose
``Family`` class mirrors the slice of the reference's ``class family`` that the path
uses (/root/reference/src/family.h:225-375): construct from PED members, ``set_LK`` /
``calPostProbBN`` / ``get_postProb`` / ``get_postProbSingle`` / ``get_postRlt`` — except
that ``set_LK`` takes a batch of sites, because the boundary is batched.

There is no CPU compute path: if the shared library is missing, or no gfx950 device is
usable, the calls raise.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

from . import shard, synth  # noqa: F401
from .pedigree import Pedigree, read_ped, synthetic_pedigree  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libfamseq_hip.so")
MAXN = 20
TEXT_STRIDE = 80  # FAMSEQ_TEXT_STRIDE

ST_OK, ST_SINGLE_FAIL, ST_BN_FAIL, ST_SHORTCUT = 0, 1, 2, 0x80
FLAG_KNOWN, FLAG_CHRX = 1, 2
ENGINE_ENUM, ENGINE_ELIM = 0, 1


class FamseqError(RuntimeError):
    pass


class CModel(C.Structure):
    """famseq_model (include/famseq_hip.h)."""
    _fields_ = [
        ("n_members", C.c_int32),
        ("mother", C.c_int32 * MAXN),
        ("father", C.c_int32 * MAXN),
        ("gender", C.c_int32 * MAXN),
        ("sequenced", C.c_uint8 * MAXN),
        ("pcp2", C.c_double * 27),
        ("pcp2Xf", C.c_double * 27),
        ("pcp2Xm", C.c_double * 27),
        ("genoProbN", C.c_double * 3),
        ("genoProbK", C.c_double * 3),
        ("genoProbXN", C.c_double * 3),
        ("genoProbXK", C.c_double * 3),
        ("lc", C.c_double),
    ]


class CPedigree(C.Structure):
    """famseq_pedigree (include/famseq_hip.h): the size-independent model; the per-member arrays are numpy arrays
    kept alive in ``_keep``."""
    _fields_ = [
        ("n_members", C.c_int32),
        ("mother", C.POINTER(C.c_int32)),
        ("father", C.POINTER(C.c_int32)),
        ("gender", C.POINTER(C.c_int32)),
        ("sequenced", C.POINTER(C.c_uint8)),
        ("pcp2", C.c_double * 27),
        ("pcp2Xf", C.c_double * 27),
        ("pcp2Xm", C.c_double * 27),
        ("genoProbN", C.c_double * 3),
        ("genoProbK", C.c_double * 3),
        ("genoProbXN", C.c_double * 3),
        ("genoProbXK", C.c_double * 3),
        ("lc", C.c_double),
    ]


# every symbol include/famseq_hip.h declares
ABI_SYMBOLS = [
    "famseq_transmission_tables", "famseq_model_init", "famseq_pedigree_init", "famseq_device_count", "famseq_create",
    "famseq_create_pedigree",
    "famseq_destroy", "famseq_last_error", "famseq_set_option", "famseq_plan_json",
    "famseq_bn_batch", "famseq_bn_batch_sharded", "famseq_bn_batch_device", "famseq_bn_batch_device_sharded",
    "famseq_bn_call_batch", "famseq_bn_call_text_batch", "famseq_bn_call_batch_device", "famseq_format_probe", "famseq_alloc_pinned", "famseq_free_pinned", "famseq_stream_probe",
    "famseq_call_genotypes",
]
PL_MISSING = 0xFFFF

_lib = None


def lib():
    """Load libfamseq_hip.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FamseqError("%s not found: run `make` (or __graft_entry__.build()) first; "
                          "there is no fallback implementation" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so, and ours is
    # linked against the unversioned name so that it binds to whichever is already loaded
    # (see Makefile).  Import torch first so a later `import torch` cannot bring a second one.
    if "torch" not in sys.modules and not os.environ.get("FAMSEQ_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    mp = C.POINTER(CModel)
    L.famseq_transmission_tables.argtypes = [C.c_double, dp, dp, dp]
    L.famseq_transmission_tables.restype = None
    L.famseq_model_init.argtypes = [mp, C.c_int32, ip, ip, ip, ip, bp, C.c_double, C.c_double]
    L.famseq_model_init.restype = C.c_int
    L.famseq_pedigree_init.argtypes = [C.POINTER(CPedigree), C.c_int32, ip, ip, ip, ip, bp, C.c_double, C.c_double, ip, ip]
    L.famseq_pedigree_init.restype = C.c_int
    L.famseq_create_pedigree.argtypes = [C.POINTER(CPedigree), C.c_int, C.c_char_p, C.c_size_t]
    L.famseq_create_pedigree.restype = C.c_void_p
    L.famseq_device_count.restype = C.c_int
    L.famseq_create.argtypes = [mp, C.c_int, C.c_char_p, C.c_size_t]
    L.famseq_create.restype = C.c_void_p
    L.famseq_destroy.argtypes = [C.c_void_p]
    L.famseq_destroy.restype = None
    L.famseq_last_error.argtypes = [C.c_void_p]
    L.famseq_last_error.restype = C.c_char_p
    L.famseq_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.famseq_set_option.restype = C.c_int
    L.famseq_plan_json.argtypes = [C.c_void_p]
    L.famseq_plan_json.restype = C.c_char_p
    L.famseq_bn_batch.argtypes = [C.c_void_p, C.c_int64, dp, bp, dp, dp, bp]
    L.famseq_bn_batch.restype = C.c_int
    L.famseq_bn_batch_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, dp, bp, dp, dp, bp]
    L.famseq_bn_batch_sharded.restype = C.c_int
    vp = C.c_void_p
    L.famseq_bn_batch_device.argtypes = [C.c_void_p, C.c_int64, vp, vp, vp, vp, vp, vp]
    L.famseq_bn_batch_device.restype = C.c_int
    pp = C.POINTER(C.c_void_p)
    L.famseq_bn_batch_device_sharded.argtypes = [pp, C.c_int, C.POINTER(C.c_int64), pp, pp, pp, pp, pp]
    L.famseq_bn_batch_device_sharded.restype = C.c_int
    L.famseq_bn_call_batch.argtypes = [C.c_void_p, C.c_int64, dp, C.POINTER(C.c_uint16), bp, ip, C.c_int32, dp, dp,
                                       C.POINTER(C.c_int8), bp]
    L.famseq_bn_call_batch.restype = C.c_int
    L.famseq_bn_call_text_batch.argtypes = [C.c_void_p, C.c_int64, dp, C.POINTER(C.c_uint16), bp, ip, C.c_int32, C.c_char_p, bp]
    L.famseq_bn_call_text_batch.restype = C.c_int
    L.famseq_bn_call_batch_device.argtypes = [C.c_void_p, C.c_int64, vp, vp, vp, ip, C.c_int32, vp, vp, vp, vp, vp, vp]
    L.famseq_bn_call_batch_device.restype = C.c_int
    L.famseq_format_probe.argtypes = [C.c_void_p, C.c_int64, dp, C.c_char_p]
    L.famseq_format_probe.restype = C.c_int
    L.famseq_stream_probe.argtypes = [C.c_void_p, C.c_int64, vp, vp, vp, vp]
    L.famseq_stream_probe.restype = C.c_int
    L.famseq_alloc_pinned.argtypes = [C.c_size_t]
    L.famseq_alloc_pinned.restype = C.c_void_p
    L.famseq_free_pinned.argtypes = [C.c_void_p]
    L.famseq_free_pinned.restype = None
    L.famseq_call_genotypes.argtypes = [dp, C.c_int64, C.POINTER(C.c_int8)]
    L.famseq_call_genotypes.restype = None
    _lib = L
    return L


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def transmission_tables(mrate):
    a, b, c = (np.zeros(27) for _ in range(3))
    lib().famseq_transmission_tables(float(mrate), _p(a, C.c_double), _p(b, C.c_double), _p(c, C.c_double))
    return a, b, c


def device_count():
    return lib().famseq_device_count()


def make_model(ped: Pedigree, mrate=1e-7, lc=1.0, genoProbN=None, genoProbK=None, genoProbXN=None,
               genoProbXK=None, sequenced=None, size_independent=False):
    """family(mem, mRate) + set_genoProb* + set_lc + init()  (file.cpp:1888-1927).  Up to 20 members: the fixed
    famseq_model; beyond (or size_independent=True): famseq_pedigree, which only the sum-product engine serves."""
    i32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
    ids, mids, fids, gen = i32(ped.ids), i32(ped.mids), i32(ped.fids), i32(ped.genders)
    seq = np.ascontiguousarray(ped.sequenced if sequenced is None else sequenced, dtype=np.uint8)
    if ped.n > MAXN or size_independent:
        m = CPedigree()
        mo, fa = np.zeros(ped.n, np.int32), np.zeros(ped.n, np.int32)
        m._keep = (gen, seq, mo, fa)  # the struct points into these
        rc = lib().famseq_pedigree_init(C.byref(m), ped.n, _p(ids, C.c_int32), _p(mids, C.c_int32), _p(fids, C.c_int32),
                                        _p(gen, C.c_int32), _p(seq, C.c_uint8), float(mrate), float(lc), _p(mo, C.c_int32),
                                        _p(fa, C.c_int32))
    else:
        m = CModel()
        rc = lib().famseq_model_init(C.byref(m), ped.n, _p(ids, C.c_int32), _p(mids, C.c_int32), _p(fids, C.c_int32),
                                     _p(gen, C.c_int32), _p(seq, C.c_uint8), float(mrate), float(lc))
    if rc != 0:
        msg = {-10: "This is not a fulfill family. Please check the ped file.",
               -11: "a mother is not female or a father is not male"}.get(rc, "bad pedigree arguments")
        raise FamseqError("famseq_model_init: %s (%d)" % (msg, rc))
    for name, v in (("genoProbN", genoProbN), ("genoProbK", genoProbK), ("genoProbXN", genoProbXN),
                    ("genoProbXK", genoProbXK)):
        if v is not None:
            for g in range(3):
                getattr(m, name)[g] = float(v[g])
    return m


class Context:
    """famseq_ctx: one model bound to one GPU (device=-1: plan only, no compute)."""

    def __init__(self, model, device=0, **options):
        self.n = model.n_members
        self._model = model
        err = C.create_string_buffer(512)
        create = lib().famseq_create_pedigree if isinstance(model, CPedigree) else lib().famseq_create
        self._h = create(C.byref(model), int(device), err, len(err))
        if not self._h:
            raise FamseqError("famseq_create: " + err.value.decode())
        for k, v in options.items():
            self.set_option(k, v)

    def close(self):
        if getattr(self, "_h", None):
            lib().famseq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise FamseqError("%s failed (%d): %s" % (what, rc, lib().famseq_last_error(self._h).decode()))

    def set_option(self, key, value):
        self._check(lib().famseq_set_option(self._h, key.encode(), int(value)), "famseq_set_option(%s)" % key)

    def plan(self):
        return json.loads(lib().famseq_plan_json(self._h).decode())

    def bn_batch(self, lk, flags=None, want_single=True, want_status=True):
        """Host arrays in, host arrays out: (post, single, status)."""
        lk = np.ascontiguousarray(lk, dtype=np.float64).reshape(-1, self.n, 3)
        s = lk.shape[0]
        fl = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint8)
        if fl is not None and fl.shape != (s,):
            raise ValueError("flags must have one byte per site")
        post = np.empty_like(lk)
        single = np.empty_like(lk) if want_single else None
        status = np.zeros(s, np.uint8) if want_status else None
        rc = lib().famseq_bn_batch(self._h, s, _p(lk, C.c_double), None if fl is None else _p(fl, C.c_uint8),
                                   _p(post, C.c_double), None if single is None else _p(single, C.c_double),
                                   None if status is None else _p(status, C.c_uint8))
        self._check(rc, "famseq_bn_batch")
        return post, single, status

    def bn_call_batch(self, seq_members, lk=None, pl16=None, flags=None):
        """Fused call path: -> (gpp[S,n_seq,3], fpp[S,n_seq,3], fgt[S,n_seq], status[S]).
        Input is either lk [S,N,3] float64 or pl16 [S,n_seq,3] uint16 (VCF column order)."""
        seq = np.ascontiguousarray(seq_members, dtype=np.int32)
        k = len(seq)
        if (lk is None) == (pl16 is None):
            raise ValueError("give exactly one of lk / pl16")
        if lk is not None:
            lk = np.ascontiguousarray(lk, dtype=np.float64).reshape(-1, self.n, 3)
            s = lk.shape[0]
        else:
            pl16 = np.ascontiguousarray(pl16, dtype=np.uint16).reshape(-1, k, 3)
            s = pl16.shape[0]
        fl = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint8)
        gpp, fpp = np.empty((s, k, 3)), np.empty((s, k, 3))
        fgt, status = np.empty((s, k), np.int8), np.zeros(s, np.uint8)
        rc = lib().famseq_bn_call_batch(self._h, s, None if lk is None else _p(lk, C.c_double),
                                        None if pl16 is None else _p(pl16, C.c_uint16),
                                        None if fl is None else _p(fl, C.c_uint8), _p(seq, C.c_int32), k,
                                        _p(gpp, C.c_double), _p(fpp, C.c_double), _p(fgt, C.c_int8), _p(status, C.c_uint8))
        self._check(rc, "famseq_bn_call_batch")
        return gpp, fpp, fgt, status

    def bn_call_text_batch(self, seq_members, lk=None, pl16=None, flags=None):
        """The call path with its outputs as text: -> (records, status).  records[s][j] is the bytes the reference's drivers
        append to sample column j of site s, b"g0,g1,g2:f0,f1,f2:0/1\\t" (file.cpp:696-745), formatted on the device."""
        seq = np.ascontiguousarray(seq_members, dtype=np.int32)
        k = len(seq)
        if (lk is None) == (pl16 is None):
            raise ValueError("give exactly one of lk / pl16")
        if lk is not None:
            lk = np.ascontiguousarray(lk, dtype=np.float64).reshape(-1, self.n, 3)
            s = lk.shape[0]
        else:
            pl16 = np.ascontiguousarray(pl16, dtype=np.uint16).reshape(-1, k, 3)
            s = pl16.shape[0]
        fl = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint8)
        text = np.zeros((s, k, TEXT_STRIDE), np.uint8)
        status = np.zeros(s, np.uint8)
        rc = lib().famseq_bn_call_text_batch(self._h, s, None if lk is None else _p(lk, C.c_double),
                                             None if pl16 is None else _p(pl16, C.c_uint16),
                                             None if fl is None else _p(fl, C.c_uint8), _p(seq, C.c_int32), k,
                                             text.ctypes.data_as(C.c_char_p), _p(status, C.c_uint8))
        self._check(rc, "famseq_bn_call_text_batch")
        return text, status

    def bn_call_batch_device(self, n_sites, seq_members, d_lk=0, d_pl16=0, d_flags=0, d_gpp=0, d_fpp=0, d_fgt=0, d_status=0, d_text=0,
                             stream=0):
        """The call path on resident buffers (raw device pointers as ints; 0 = not given); enqueues on `stream` and returns."""
        seq = np.ascontiguousarray(seq_members, dtype=np.int32)
        rc = lib().famseq_bn_call_batch_device(self._h, int(n_sites), d_lk or None, d_pl16 or None, d_flags or None, _p(seq, C.c_int32),
                                               len(seq), d_gpp or None, d_fpp or None, d_fgt or None, d_status or None, d_text or None,
                                               stream or None)
        self._check(rc, "famseq_bn_call_batch_device")

    def g6_probe(self, values):
        """The device formatter alone (famseq_format_probe): -> list of bytes, one per value."""
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        out = np.zeros((len(v), 16), np.uint8)
        self._check(lib().famseq_format_probe(self._h, len(v), _p(v, C.c_double), out.ctypes.data_as(C.c_char_p)), "famseq_format_probe")
        return [bytes(r[:r[15]]) for r in out]

    def bn_batch_device(self, n_sites, d_lk, d_flags, d_post, d_single=0, d_status=0, stream=0):
        """Raw device pointers (ints); enqueues on `stream` and returns."""
        rc = lib().famseq_bn_batch_device(self._h, int(n_sites), d_lk, d_flags or None, d_post, d_single or None,
                                          d_status or None, stream or None)
        self._check(rc, "famseq_bn_batch_device")


def stream_probe(ctx, n_doubles, d_in, d_out1, d_out2, stream=0):
    """famseq_stream_probe: the kernels' traffic shape as a bare elementwise kernel (diagnostic)."""
    ctx._check(lib().famseq_stream_probe(ctx._h, int(n_doubles), d_in, d_out1, d_out2, stream or None), "famseq_stream_probe")


def bn_batch_sharded(contexts, lk, flags=None):
    """famseq_bn_batch over several contexts (one per GPU) from this process: contiguous site ranges,
    one host thread per context.  -> (post, single, status)."""
    n = contexts[0].n
    lk = np.ascontiguousarray(lk, dtype=np.float64).reshape(-1, n, 3)
    s = lk.shape[0]
    fl = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint8)
    post, single, status = np.empty_like(lk), np.empty_like(lk), np.zeros(s, np.uint8)
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    rc = lib().famseq_bn_batch_sharded(arr, len(contexts), s, _p(lk, C.c_double), None if fl is None else _p(fl, C.c_uint8),
                                       _p(post, C.c_double), _p(single, C.c_double), _p(status, C.c_uint8))
    if rc != 0:
        msgs = [lib().famseq_last_error(c._h).decode() for c in contexts]
        raise FamseqError("famseq_bn_batch_sharded failed (%d): %s" % (rc, "; ".join(m for m in msgs if m)))
    return post, single, status


def bn_batch_device_sharded(contexts, n_sites, d_lk, d_flags, d_post, d_single=None, d_status=None):
    """famseq_bn_batch_device_sharded: shard g = n_sites[g] sites resident on contexts[g]'s device, given as
    raw device pointers (ints; 0 / None = absent).  Blocking; one host thread per context."""
    g = len(contexts)

    def ptrs(v):
        if v is None:
            return None
        return (C.c_void_p * g)(*[(x or None) for x in v])

    arr = (C.c_void_p * g)(*[c._h for c in contexts])
    ns = (C.c_int64 * g)(*[int(x) for x in n_sites])
    rc = lib().famseq_bn_batch_device_sharded(arr, g, ns, ptrs(d_lk), ptrs(d_flags), ptrs(d_post), ptrs(d_single),
                                              ptrs(d_status))
    if rc != 0:
        msgs = [lib().famseq_last_error(c._h).decode() for c in contexts]
        raise FamseqError("famseq_bn_batch_device_sharded failed (%d): %s" % (rc, "; ".join(m for m in msgs if m)))


def call_genotypes(post):
    post = np.ascontiguousarray(post, dtype=np.float64).reshape(-1, 3)
    out = np.empty(post.shape[0], np.int8)
    lib().famseq_call_genotypes(_p(post, C.c_double), post.shape[0], _p(out, C.c_int8))
    return out


class Family:
    """Batched mirror of the reference's `family` for the BN path.

    ref = family(mem, mRate); ref.set_lc(lc); ref.init(); ref.set_mapV2P(...)
    then per site set_LK(lk) / calPostProbBN(Known, chrType) / get_postProb() ...
    Here set_LK takes [S,N,3] (PED order) and calPostProbBN takes per-site arrays."""

    def __init__(self, ped: Pedigree, mrate=1e-7, lc=1.0, device=0, **priors):
        self.ped = ped
        self.model = make_model(ped, mrate, lc, **priors)
        self.ctx = Context(self.model, device)
        self._seq_idx = np.nonzero(ped.sequenced)[0]
        self._lk = None
        self._post = self._single = self._status = None

    def get_numInd(self):
        return self.ped.n

    def get_realNumInd(self):
        return len(self._seq_idx)

    def set_mapV2P(self, v2p):
        """VCF column -> PED index (or -1); fixes the order of the get_* rows (family.cpp:364-376)."""
        self._seq_idx = np.array([p for p in v2p if p >= 0], dtype=np.int64)

    def set_LK(self, lk):
        lk = np.asarray(lk, dtype=np.float64)
        if lk.ndim == 2:
            lk = lk[None]
        if lk.shape[1:] != (self.ped.n, 3):
            print("The dimention of likelihood matrix is wrong. Cannot set likelihood.")
            return False
        self._lk = lk
        self._post = self._single = self._status = None
        return True

    def calPostProbBN(self, Known=False, chrType=0):
        """-> bool array per site (True where the reference would return true)."""
        if self._lk is None:
            print("Likelihood has not been set. Please set likelihood first.")
            return False
        s = self._lk.shape[0]
        flags = (np.broadcast_to(np.asarray(Known, dtype=bool), (s,)).astype(np.uint8) * FLAG_KNOWN
                 | np.broadcast_to(np.asarray(chrType) == 1, (s,)).astype(np.uint8) * FLAG_CHRX)
        self._post, self._single, self._status = self.ctx.bn_batch(self._lk, flags)
        return (self._status & 3) == 0

    def get_status(self):
        return self._status

    def get_postProb(self, flag=True):
        return self._post[:, self._seq_idx, :] if flag else self._post

    def get_postProbSingle(self, flag=True):
        return self._single[:, self._seq_idx, :] if flag else self._single

    def get_postRlt(self):
        p = self.get_postProb(True)
        return call_genotypes(p).reshape(p.shape[0], p.shape[1])
