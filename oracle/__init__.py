"""ctypes front-end for the CPU oracle — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``famseq_amd``) never does.

* ``OracleModel``  wraps ``liboracle_bn.so`` (oracle/bn_oracle.c, our plain-C
  restatement of /root/reference/src/family.cpp:750-1124 and friends).
* ``RefFamily``    wraps ``_ref/libfamseq_ref.so`` (the *compiled reference*
  behind oracle/ref_harness.cpp).  Exists only where ``make -C oracle ref`` ran
  (needs /root/reference); used to pin the restatement and to generate the
  fixtures in tests/golden/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAXN = 20

ST_OK, ST_SINGLE_FAIL, ST_BN_FAIL, ST_SHORTCUT = 0, 1, 2, 0x80


class _CModel(C.Structure):
    _fields_ = [
        ("n", C.c_int32),
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


def build(ref=None):
    """Compile liboracle_bn.so (and _ref/ when the reference tree is present)."""
    targets = ["all"]
    if ref is None:
        ref = os.path.isdir("/root/reference/src")
    if ref:
        targets.append("ref")
    subprocess.check_call(["make", "-s", "-C", HERE] + targets)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle_bn.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        L.oracle_tables.argtypes = [C.c_double, dp, dp, dp]
        L.oracle_tables.restype = None
        L.oracle_model_init.argtypes = [C.POINTER(_CModel), C.c_int, ip, ip, ip, ip, bp, C.c_double, C.c_double]
        L.oracle_model_init.restype = C.c_int
        L.oracle_bn_site.argtypes = [C.POINTER(_CModel), dp, C.c_int, C.c_int, dp, dp]
        L.oracle_bn_site.restype = C.c_uint8
        L.oracle_bn_batch.argtypes = [C.POINTER(_CModel), C.c_int64, dp, bp, dp, dp, bp, C.c_int]
        L.oracle_bn_batch.restype = None
        L.oracle_argmax3.argtypes = [dp]
        L.oracle_argmax3.restype = C.c_int
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _i32(x):
    return np.ascontiguousarray(x, dtype=np.int32)


def tables(mu):
    a, b, c = (np.zeros(27) for _ in range(3))
    lib().oracle_tables(float(mu), _p(a, C.c_double), _p(b, C.c_double), _p(c, C.c_double))
    return a, b, c


class OracleModel:
    """Pedigree + constants; ``ids/mids/fids/genders`` are the PED columns."""

    def __init__(self, ids, mids, fids, genders, sequenced=None, mrate=1e-7, lc=1.0,
                 genoProbN=None, genoProbK=None, genoProbXN=None, genoProbXK=None):
        n = len(ids)
        self.n = n
        seq = np.ones(n, np.uint8) if sequenced is None else np.ascontiguousarray(sequenced, dtype=np.uint8)
        self.c = _CModel()
        a, b, c_, d = _i32(ids), _i32(mids), _i32(fids), _i32(genders)
        rc = lib().oracle_model_init(C.byref(self.c), n, _p(a, C.c_int32), _p(b, C.c_int32),
                                     _p(c_, C.c_int32), _p(d, C.c_int32), _p(seq, C.c_uint8),
                                     float(mrate), float(lc))
        if rc != 0:
            raise ValueError("oracle_model_init failed: %d" % rc)
        for name, v in (("genoProbN", genoProbN), ("genoProbK", genoProbK),
                        ("genoProbXN", genoProbXN), ("genoProbXK", genoProbXK)):
            if v is not None:
                for g in range(3):
                    getattr(self.c, name)[g] = float(v[g])

    @property
    def mother(self):
        return np.array(self.c.mother[: self.n], dtype=np.int32)

    @property
    def father(self):
        return np.array(self.c.father[: self.n], dtype=np.int32)

    def table(self, name):
        return np.array(getattr(self.c, name)[:], dtype=np.float64)

    def bn_batch(self, lk, flags=None, threads=1):
        lk = np.ascontiguousarray(lk, dtype=np.float64).reshape(-1, self.n, 3)
        s = lk.shape[0]
        fl = np.zeros(s, np.uint8) if flags is None else np.ascontiguousarray(flags, dtype=np.uint8)
        post, single = np.empty_like(lk), np.empty_like(lk)
        st = np.zeros(s, np.uint8)
        lib().oracle_bn_batch(C.byref(self.c), s, _p(lk, C.c_double), _p(fl, C.c_uint8),
                              _p(post, C.c_double), _p(single, C.c_double), _p(st, C.c_uint8), int(threads))
        return post, single, st


def argmax3(row):
    r = np.ascontiguousarray(row, dtype=np.float64)
    return lib().oracle_argmax3(_p(r, C.c_double))


# --------------------------------------------------------------------------
# compiled reference (only where oracle/_ref was built)
# --------------------------------------------------------------------------
REF_SO = os.path.join(HERE, "_ref", "libfamseq_ref.so")
REF_CLI = os.path.join(HERE, "_ref", "FamSeq_ref")


def have_ref():
    return os.path.exists(REF_SO)


_rlib = None


def _ref():
    global _rlib
    if _rlib is None:
        L = C.CDLL(REF_SO)
        dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        L.famref_create.argtypes = [C.c_int, ip, ip, ip, ip, bp, C.c_double, C.c_double, dp, dp, dp, dp]
        L.famref_create.restype = C.c_void_p
        L.famref_destroy.argtypes = [C.c_void_p]
        L.famref_tables.argtypes = [C.c_void_p, dp, dp, dp]
        L.famref_site.argtypes = [C.c_void_p, C.c_int, dp, C.c_int, C.c_int, dp, dp]
        L.famref_site.restype = C.c_int
        _rlib = L
    return _rlib


class RefFamily:
    """The reference's own `class family`, driven through ref_harness.cpp."""

    def __init__(self, ids, mids, fids, genders, sequenced=None, mrate=1e-7, lc=1.0,
                 genoProbN=None, genoProbK=None, genoProbXN=None, genoProbXK=None):
        n = len(ids)
        self.n = n
        seq = np.ones(n, np.uint8) if sequenced is None else np.ascontiguousarray(sequenced, dtype=np.uint8)
        arrs = [_i32(ids), _i32(mids), _i32(fids), _i32(genders)]

        def opt(v):
            if v is None:
                return None
            return _p(np.ascontiguousarray(v, dtype=np.float64), C.c_double)

        self._keep = [np.ascontiguousarray(v, dtype=np.float64) for v in (genoProbN, genoProbK, genoProbXN, genoProbXK) if v is not None]
        self.h = _ref().famref_create(n, *[_p(a, C.c_int32) for a in arrs], _p(seq, C.c_uint8),
                                      float(mrate), float(lc), opt(genoProbN), opt(genoProbK),
                                      opt(genoProbXN), opt(genoProbXK))
        if not self.h:
            raise ValueError("reference family::init() rejected the pedigree")

    def tables(self):
        a, b, c = (np.zeros(27) for _ in range(3))
        _ref().famref_tables(self.h, _p(a, C.c_double), _p(b, C.c_double), _p(c, C.c_double))
        return a, b, c

    def bn_batch(self, lk, flags=None, method=1):
        lk = np.ascontiguousarray(lk, dtype=np.float64).reshape(-1, self.n, 3)
        s = lk.shape[0]
        fl = np.zeros(s, np.uint8) if flags is None else np.asarray(flags, dtype=np.uint8)
        post, single = np.empty_like(lk), np.empty_like(lk)
        st = np.zeros(s, np.uint8)
        for i in range(s):
            st[i] = _ref().famref_site(self.h, method, _p(lk[i], C.c_double), int(fl[i] & 1), int((fl[i] >> 1) & 1),
                                       _p(post[i], C.c_double), _p(single[i], C.c_double))
        return post, single, st

    def __del__(self):
        try:
            if self.h:
                _ref().famref_destroy(self.h)
                self.h = None
        except Exception:
            pass
