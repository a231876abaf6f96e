"""The generated kernels' arithmetic, checked without a GPU.

The per-pedigree kernels (csrc/enum_codegen.cpp, csrc/elim_codegen.cpp) are straight-line C++ inside
a HIP shell.  Generated for a one-lane workgroup (FAMSEQ_*_BT=1) the shell degenerates to a plain
loop over sites, so the very same source — HIP qualifiers defined away, compiler pins and barriers
removed — compiles with g++ and can be compared with the oracle here.  This is a test of the
GENERATORS (indexing, tables, accumulation order, status rules); it is not a product path: nothing
in famseq_amd/ can run this way, and the GPU parity tests remain the check of what ships.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import famseq_amd as fs
from _cases import load_cases

SHIM = r"""
#include <cmath>
struct v2d { double x, y; };
struct idx3_ { int x; };
static idx3_ threadIdx{0}, blockIdx{0}, gridDim{1};
#define __global__
#define __shared__ static
#define __launch_bounds__(...)
#define __builtin_amdgcn_ballot_w64(p) ((unsigned long)(p))
#define __builtin_amdgcn_readfirstlane(v) (v)
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
"""


# the call-path forms use three device builtins and __umulhi: host equivalents
SHIM_CALL = r"""
#include <cmath>
#define FS_RCP(x) (1.0 / (x))
struct fs_v2d { double x, y; };
#define FS_IS_POS_FINITE(x) ((x) > 0 && (x) < INFINITY)
#define FS_KEEP_BRANCH() (void)0
#define FS_GLOBAL
static inline double fs_mant_(double x) { int e; return std::frexp(x, &e); }
static inline int fs_exp_(double x) { int e; std::frexp(x, &e); return e; }
#define FS_FREXP_MANT(x) fs_mant_(x)
#define FS_FREXP_EXP(x) fs_exp_(x)
#define FS_UMULHI(a, b) ((unsigned)(((unsigned long long)(a) * (unsigned long long)(b)) >> 32))
static inline int fs_hi32_(double x) { unsigned long long u; __builtin_memcpy(&u, &x, 8); return (int)(u >> 32); }
#define FS_HI32(x) fs_hi32_(x)
#define __device__
#define __forceinline__ inline
"""

# Lanes-per-site kernels need their G lanes to meet at the workgroup barriers: the lanes are host
# threads here (thread-local threadIdx, the `static` LDS arrays shared), the barrier a pthread barrier.
SHIM_THREADS = r"""
#include <cmath>
#include <pthread.h>
#include <thread>
#include <vector>
struct v2d { double x, y; };
struct idx3_ { int x; };
static thread_local idx3_ threadIdx{0};
static idx3_ blockIdx{0}, gridDim{1};
static pthread_barrier_t wg_barrier_;
#define __global__
#define __shared__ static
#define __launch_bounds__(...)
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define famseq_enum_lane famseq_enum_lane_one_thread
"""
DRIVER_THREADS = r"""
#undef famseq_enum_lane
extern "C" void famseq_enum_lane(const double *lk, const unsigned char *fl, double *post, double *single,
                                 unsigned char *st, long n, const double *tc, double lc) {
  pthread_barrier_init(&wg_barrier_, nullptr, BT);
  std::vector<std::thread> lanes;
  for (int t = 0; t < BT; ++t)
    lanes.emplace_back([=] { threadIdx.x = t; famseq_enum_lane_one_thread(lk, fl, post, single, st, n, tc, lc); });
  for (auto &l : lanes) l.join();
  pthread_barrier_destroy(&wg_barrier_);
}
"""


# The single posterior's quotients: on the device three divisions share one refined reciprocal where every intermediate is a normal
# number (csrc/elim_codegen.cpp kDiv3Text: the compiler's own division sequence, bit for bit); a host build takes the plain divisions.
HOST_DIV = "#define FS_DIV_OK(p0, p1, p2, s) 0\n#define FS_DIV3_FAST(p0, p1, p2, s, o0, o1, o2) (void)0\n"


def portable(src: str) -> str:
    """Device-only spellings both shims replace: compiler pins (no arithmetic in them) and clang's vector types."""
    src = src.replace("#pragma clang fp contract(off)", "#pragma clang fp contract(off)\n" + HOST_DIV, 1)
    src = re.sub(r"#define FS_HIDE_LANE\(t_\).*", "#define FS_HIDE_LANE(t_) (void)0", src)
    src = src.replace("typedef unsigned fs_v4u __attribute__((ext_vector_type(4)));", "struct fs_v4u { unsigned x, y, z, w; };")
    return src.replace("__builtin_amdgcn_sched_barrier(0);", "")


def host_source(src: str, threads=False) -> str:
    src = re.sub(r"#define FS_KEEP_BRANCH\(\) asm.*", "", src)  # device default; the host shim has its own
    src = src.replace("typedef double fs_v2d __attribute__((ext_vector_type(2)));", "")
    src = portable(src)
    if threads:
        src = src.replace("#include <hip/hip_runtime.h>", SHIM_THREADS)
        src = re.sub(r"#define LDS_BARRIER\(\).*", "#define LDS_BARRIER() pthread_barrier_wait(&wg_barrier_)", src)
        src = src.replace('extern "C" __global__', "static") + DRIVER_THREADS
        src = re.sub(r'asm volatile\(""[^\n;]*\);', "", src)
        assert "asm" not in src
        return src
    src = src.replace("#include <hip/hip_runtime.h>", SHIM + (SHIM_CALL if "fs_call_args" in src else ""))
    src = re.sub(r"#define LDS_BARRIER\(\).*", "#define LDS_BARRIER() (void)0", src)
    src = src.replace("typedef double v2d __attribute__((ext_vector_type(2)));", "")
    src = src.replace("__attribute__((address_space(3)))", "").replace("__attribute__((address_space(1)))", "")
    src = re.sub(r'asm volatile\(""[^\n;]*\);', "", src)  # compiler pins / fences: no arithmetic in them
    assert "asm" not in src
    return src


def factor_tables(m):
    """Tc[4 flag combos][founder male, founder female, child male, child female][27] — the block
    csrc/bn_kernel.hip build_factor_tables uploads (family.cpp:885-893, :992-1009, :1052-1073)."""
    tc = np.zeros((4, 4, 27))
    for fl in range(4):
        known, x = fl & 1, fl & 2
        autos = np.array(m.genoProbK[:] if known else m.genoProbN[:])
        male = np.array((m.genoProbXK[:] if known else m.genoProbXN[:])) if x else autos
        tc[fl, 0, 0:27:9], tc[fl, 1, 0:27:9] = male, autos
        tc[fl, 2] = m.pcp2Xm[:] if x else m.pcp2[:]
        tc[fl, 3] = m.pcp2Xf[:] if x else m.pcp2[:]
    return tc.reshape(-1)


def misaligned(shape, dtype=np.float64):
    """An array whose data pointer is 8 mod 16: the generated code then takes its 8-byte staging
    path (the 16-byte one steps by BT / 2 lanes, which a one-lane block does not have)."""
    n = int(np.prod(shape))
    raw = np.zeros(n + 2, dtype=dtype)
    off = 1 if raw.ctypes.data % 16 == 0 else 0
    out = raw[off:off + n].reshape(shape)
    assert out.ctypes.data % 16 == 8
    return out


def run_host(fn, model, lk, flags, lc=None):
    n_sites, n = lk.shape[0], lk.shape[1]
    a = misaligned(lk.shape)
    a[...] = lk
    post, single = misaligned(lk.shape), misaligned(lk.shape)
    post[...] = -1
    single[...] = -1
    st = np.full(n_sites, 77, np.uint8)
    fl = np.ascontiguousarray(flags, np.uint8)
    tc = np.ascontiguousarray(factor_tables(model))
    fn(a.ctypes.data, fl.ctypes.data, post.ctypes.data, single.ctypes.data, st.ctypes.data, n_sites, tc.ctypes.data,
       float(model.lc if lc is None else lc))
    return post, single, st


CASES = load_cases()


@pytest.mark.parametrize("kind", ["lane", "elim"])
@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_generated_arithmetic_matches_the_fixtures(case, kind, tmp_path, monkeypatch):
    ped = case.pedigree()
    model = fs.make_model(ped, **case.consts)
    probe = fs.Context(model, device=-1)
    supported = probe.plan()["elim_supported"]
    probe.close()
    if kind == "elim" and not supported:
        pytest.skip("more loops than the sum-product engine conditions on")
    fn = build_host_kernel(model, kind, tmp_path, monkeypatch)
    post, single, st = run_host(fn, model, case.lk, case.flags)
    assert np.array_equal(st, case.status)
    ok = (case.status & 3) == 0
    assert np.array_equal(single[(case.status & 3) != 1], case.single[(case.status & 3) != 1])
    np.testing.assert_allclose(post[ok], case.post[ok], rtol=1e-12, atol=0)
    assert np.all(np.isnan(post[~ok]))


def build_host_kernel(model, kind, tmp_path, monkeypatch):
    """Generate the kernel of `model` for a one-lane workgroup and compile its source for the host."""
    cache = tmp_path / ("cache_" + kind)
    cache.mkdir(exist_ok=True)
    for k, v in dict(FAMSEQ_KERNEL_CACHE=str(cache), FAMSEQ_KEEP_SRC="1", FAMSEQ_LANE_BT="1", FAMSEQ_ELIM_BT="1",
                     FAMSEQ_LANE_MINWAVES="1", FAMSEQ_JIT_SOURCE_ONLY="1").items():  # the source is what is tested: no hipcc
        monkeypatch.setenv(k, v)
    ctx = fs.Context(model, device=-1)
    if kind.startswith("group"):  # lanes-per-site mode: "group<d>" = 3^d lanes per site, two sites per workgroup
        d = int(kind[5:])
        monkeypatch.setenv("FAMSEQ_LANE_BT", str(2 * 3 ** d + 1))  # + one lane that belongs to no group
        ctx.close()
        ctx = fs.Context(model, device=-1)
        ctx.set_option("group_digits", d)
        obj, entry = ctx.plan()["enum_group_code_objects"][d - 1], "famseq_enum_lane"
        ctx.close()
        src = open(obj[:-6] + ".hip").read()
        assert "#define G %d\n" % 3 ** d in src
        cpp, so = str(cache / "k.cpp"), str(cache / "k.so")
        open(cpp, "w").write(host_source(src, threads=True))
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-w", "-shared", "-fPIC", "-pthread", "-o", so, cpp])
        fn = getattr(C.CDLL(so), entry)
        fn.restype = None
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_double]
        return fn
    if kind in ("lane_call", "elim_call"):  # the fused call-path forms
        ctx.set_option("call_kernels", 1)
        plan = ctx.plan()
        obj, entry = ((plan["enum_lane_call_code_object"], "famseq_enum_lane") if kind == "lane_call"
                      else (plan["elim_call_code_object"], "famseq_elim"))
        ctx.close()
        src = open(obj[:-6] + ".hip").read()
        assert "call path" in src.splitlines()[0] and "#define BT 1\n" in src
        cpp, so = str(cache / "k.cpp"), str(cache / "k.so")
        open(cpp, "w").write(host_source(src))
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-w", "-shared", "-fPIC", "-o", so, cpp])
        fn = getattr(C.CDLL(so), entry)
        fn.restype = None
        fn.argtypes = [C.c_void_p] * 5 + [C.c_long, C.c_void_p, C.c_double, C.c_void_p]
        return fn
    if kind == "lane":
        ctx.set_option("enum_impl", 1)
        obj, entry = ctx.plan()["enum_lane_code_object"], "famseq_enum_lane"
    else:
        ctx.set_option("engine", fs.ENGINE_ELIM)
        obj, entry = ctx.plan()["elim_code_object"], "famseq_elim"
    ctx.close()
    src = open(obj[:-6] + ".hip").read()
    assert "#define BT 1\n" in src
    cpp, so = str(cache / "k.cpp"), str(cache / "k.so")
    open(cpp, "w").write(host_source(src))
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-w", "-shared", "-fPIC", "-o", so, cpp])
    fn = getattr(C.CDLL(so), entry)
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_double]
    return fn


@pytest.mark.parametrize("kind", ["lane", "elim"])
@pytest.mark.parametrize("seed", [0, 1, 2, 5])
def test_generated_arithmetic_on_random_pedigrees(seed, kind, tmp_path, monkeypatch):
    """Randomly grown pedigrees (marriage loops on seed 0, unsequenced members, hard zeros, mu = 0,
    every flag combination) against the oracle: the generators' ordering and table logic on shapes
    no fixture has."""
    import oracle
    from famseq_amd.prebuild_sets import random_pedigree
    from test_gpu_random_pedigrees import random_likelihoods

    rng, ped = random_pedigree(seed)  # famseq_amd/prebuild_sets.py: build() pre-compiles these pedigrees' kernels
    ped.relations()
    mu = [1e-7, 1e-7, 1e-4, 0.0][seed % 4]
    lk, flags = random_likelihoods(rng, ped, 48)
    ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders, ped.sequenced, mrate=mu).bn_batch(lk, flags, threads=4)
    model = fs.make_model(ped, mrate=mu)
    probe = fs.Context(model, device=-1)
    supported = probe.plan()["elim_supported"]
    probe.close()
    if kind == "elim" and not supported:
        pytest.skip("more loops than the sum-product engine conditions on")
    fn = build_host_kernel(model, kind, tmp_path, monkeypatch)
    post, single, st = run_host(fn, model, lk, flags)
    assert np.array_equal(st, ref[2])
    ok, s_ok = (st & 3) == 0, (st & 3) != 1
    assert np.array_equal(single[s_ok], ref[1][s_ok])
    np.testing.assert_allclose(post[ok], ref[0][ok], rtol=1e-10, atol=0)


def test_conditioning_on_a_first_cousin_marriage(tmp_path, monkeypatch):
    """The sum-product kernel on a pedigree with a loop (one conditioned member, three passes)
    against the oracle's enumeration, all flag combinations."""
    import oracle
    from test_elim import cousins_marry
    from test_gpu_random_pedigrees import random_likelihoods

    ped = cousins_marry()
    ped.relations()
    rng = np.random.RandomState(7)
    lk, flags = random_likelihoods(rng, ped, 64)
    ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders, ped.sequenced).bn_batch(lk, flags, threads=4)
    model = fs.make_model(ped)
    fn = build_host_kernel(model, "elim", tmp_path, monkeypatch)
    post, single, st = run_host(fn, model, lk, flags)
    assert np.array_equal(st, ref[2])
    ok = (st & 3) == 0
    assert ok.sum() > 20
    np.testing.assert_allclose(post[ok], ref[0][ok], rtol=1e-10, atol=0)


GROUP_CASES = [c for c in CASES if c.name in ("bn_synth:ped10", "bn_synth:ped10_x", "bn_synth:chain7", "bn_vcf:fam01", "bn_synth:quad_mu0")]


@pytest.mark.parametrize("d", [1, 2, 4])
@pytest.mark.parametrize("case", GROUP_CASES, ids=[c.name for c in GROUP_CASES])
def test_lanes_per_site_mode_matches_the_fixtures(case, d, tmp_path, monkeypatch):
    """The small-batch form of the enumeration kernel (3^d lanes share a site, each walking one
    combination of the d outermost looped members' digits; partial marginals summed through LDS by
    the group's first lane): the generated source with its lanes as host threads and the workgroup
    barrier as a pthread barrier, against the fixtures — ragged last chunk and an idle lane included."""
    ped = case.pedigree()
    model = fs.make_model(ped, **case.consts)
    probe = fs.Context(model, device=-1)
    dmax = probe.plan()["enum_group_digits_max"]
    probe.close()
    if d > dmax:
        pytest.skip("this pedigree's enumeration has %d looped member(s)" % dmax)
    fn = build_host_kernel(model, "group%d" % d, tmp_path, monkeypatch)
    n = min(len(case.lk), 7)  # 2 sites per chunk: three whole chunks and a ragged one
    post, single, st = run_host(fn, model, case.lk[:n], case.flags[:n])
    assert np.array_equal(st, case.status[:n])
    ok, s_ok = (case.status[:n] & 3) == 0, (case.status[:n] & 3) != 1
    assert np.array_equal(single[s_ok], case.single[:n][s_ok])
    np.testing.assert_allclose(post[ok], case.post[:n][ok], rtol=1e-12, atol=0)
    assert np.all(np.isnan(post[~ok]))


def output_slots(col, n, k):
    """member -> slot of the call-path kernels' output row: its VCF column, or — no column — one behind the k columns
    (what capi.cpp set_sequenced uploads)."""
    slot = np.array(col[:n], np.int32)
    nxt = k
    for p in range(n):
        if slot[p] < 0:
            slot[p] = nxt
            nxt += 1
    return slot


class CallArgs(C.Structure):  # struct fs_call_args of the generated source
    _fields_ = [("pl", C.c_void_p), ("lut", C.c_void_p), ("col", C.c_void_p), ("slot", C.c_void_p), ("gpp", C.c_void_p),
                ("fpp", C.c_void_p), ("fgt", C.c_void_p), ("n_seq", C.c_int32), ("magic_w", C.c_uint32), ("magic_n", C.c_uint32)]


def host_phred(p):
    with np.errstate(divide="ignore", invalid="ignore"):
        q = -10 * np.log10(p)
    return np.where(np.isposinf(q), 99999.0, np.abs(q))


CALL_CASES = [c for c in CASES if c.name in ("bn_vcf:fam01", "bn_synth:quad_mu0", "bn_synth:ped10_x", "bn_synth:ped5", "bn_lk:fam06")]


@pytest.mark.parametrize("kind", ["lane_call", "elim_call"])
@pytest.mark.parametrize("case", CALL_CASES, ids=[c.name for c in CALL_CASES])
def test_call_path_forms_match_the_fixtures(case, kind, tmp_path, monkeypatch):
    """The fused call-path form of both generated kernels on the host: fp64 rows in -> GPP / FPP / FGT in a
    shuffled VCF column order, against fabs(-10 log10) of the fixtures' posteriors (failed sites: NaN and
    -1); then the same sites as packed integer PLs (with a missing sample and a PL beyond the table)
    against the fp64 input of the table's values: identical bits."""
    import math

    ped = case.pedigree()
    model = fs.make_model(ped, **case.consts)
    if kind == "elim_call":
        probe = fs.Context(model, device=-1)
        ok_ = probe.plan()["elim_supported"]
        probe.close()
        if not ok_:
            pytest.skip("more loops than the sum-product engine conditions on")
    fn = build_host_kernel(model, kind, tmp_path, monkeypatch)
    n = ped.n
    seq = np.nonzero(case.sequenced)[0][::-1].astype(np.int32).copy()
    k = len(seq)
    col = np.full(20, -1, np.int32)
    col[seq] = np.arange(k)
    slot = output_slots(col, n, k)
    lut = np.array([math.pow(10.0, -i / 10.0) for i in range(4096)])
    tc = np.ascontiguousarray(factor_tables(model))

    def run(lk=None, pl=None, flags=None):
        S = len(flags)
        gpp, fpp = np.full((S, k, 3), -7.0), np.full((S, k, 3), -7.0)
        fgt, st = np.full((S, k), 9, np.int8), np.full(S, 77, np.uint8)
        a = CallArgs(None if pl is None else pl.ctypes.data, lut.ctypes.data, col.ctypes.data, slot.ctypes.data, gpp.ctypes.data,
                     fpp.ctypes.data, fgt.ctypes.data, k, 0xFFFFFFFF // (3 * k) + 1, 0xFFFFFFFF // k + 1)
        keep = misaligned(lk.shape) if lk is not None else None
        if lk is not None:
            keep[...] = lk
        fn(None if lk is None else keep.ctypes.data, flags.ctypes.data, None, None, st.ctypes.data, S, tc.ctypes.data,
           float(model.lc), C.addressof(a))
        return gpp, fpp, fgt, st

    flags = np.ascontiguousarray(case.flags, np.uint8)
    gpp, fpp, fgt, st = run(lk=case.lk, flags=flags)
    assert np.array_equal(st, case.status)
    ok, s_ok = (st & 3) == 0, (st & 3) != 1
    np.testing.assert_allclose(gpp[s_ok], host_phred(case.single[s_ok][:, seq]), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(fpp[ok], host_phred(case.post[ok][:, seq]), rtol=1e-9, atol=1e-9)
    assert np.array_equal(fgt[ok], fs.call_genotypes(case.post[ok][:, seq]).reshape(-1, k))
    assert np.all(np.isnan(fpp[~ok])) and np.all(fgt[~ok] == -1) and np.all(np.isnan(gpp[~s_ok]))
    if kind == "lane_call" and "lg[" in open(tmp_path / ("cache_" + kind) / "k.cpp").read():
        return  # this pedigree's lane kernel re-reads fp64 rows from global memory: the library feeds it fp64 rows only
    rng = np.random.RandomState(3)
    S = 40
    pl = rng.randint(0, 300, size=(S, k, 3)).astype(np.uint16)
    pl[np.arange(S)[:, None], np.arange(k)[None, :], rng.randint(0, 3, size=(S, k))] = 0
    pl[::5, 0] = 0xFFFF
    pl[::7, k - 1] = [0, 5000, 65534]
    table = np.append(lut, 0.0)
    lk2 = np.ones((S, n, 3))
    for j, mbr in enumerate(seq):
        v = table[np.minimum(pl[:, j].astype(np.int64), 4096)]
        v[np.all(pl[:, j] == 0xFFFF, axis=1)] = 1.0
        lk2[:, mbr] = v
    fl2 = rng.randint(0, 4, S).astype(np.uint8)
    a, b = run(lk=lk2, flags=fl2), run(pl=pl, flags=fl2)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)


def test_packed_input_is_refused_to_a_kernel_that_rereads_fp64_rows(tmp_path, monkeypatch):
    """The call-path form of a wide pedigree's lane kernel can have less LDS room than the plain form and
    then re-reads some members' likelihoods from the fp64 rows in global memory (`lg[...]`).  Such a kernel
    must never be fed packed PLs (there are no fp64 rows then: a null pointer on the GPU).  The library's
    answer to "does it re-read" has to describe the CALL form — soak seed 10 is a pedigree where the two
    forms differ (found as a GPU fault in round 2)."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _soak import soak_pedigree

    monkeypatch.setenv("FAMSEQ_KERNEL_CACHE", str(tmp_path))
    monkeypatch.setenv("FAMSEQ_KEEP_SRC", "1")
    monkeypatch.setenv("FAMSEQ_JIT_SOURCE_ONLY", "1")
    monkeypatch.setenv("FAMSEQ_LANE_CAP", "6")  # the 6-member block: the form these pedigrees ran in (the 7-member one spilled)

    def rereads(bt):
        if bt:
            monkeypatch.setenv("FAMSEQ_LANE_BT", bt)
        seen = set()
        for seed in (2, 8, 10):
            _, ped, mu = soak_pedigree(seed)
            ctx = fs.Context(fs.make_model(ped, mrate=mu), device=-1)
            ctx.set_option("enum_impl", 1)
            ctx.set_option("call_kernels", 1)
            plan = ctx.plan()
            ctx.close()
            call_src = open(plan["enum_lane_call_code_object"][:-6] + ".hip").read()
            assert plan["enum_lane_call_reads_rows"] == int("lg[" in call_src), seed
            seen.add(plan["enum_lane_call_reads_rows"])
        return seen

    # one-wave workgroups (the default since round 2's last third): a quarter of the lanes per CU, so the LDS row
    # always has room and nothing is re-read; the 256-lane form these pedigrees ran in is the one with both answers
    assert rereads(None) == {0}
    assert rereads("256") == {0, 1}


def _threaded_call_kernel(model, elim, bt, cache, monkeypatch):
    """Call-path form generated for `bt` lanes per workgroup, its lanes as host threads (the workgroup
    barrier a pthread barrier, LDS the shared statics)."""
    for k, v in dict(FAMSEQ_KERNEL_CACHE=str(cache), FAMSEQ_KEEP_SRC="1", FAMSEQ_LANE_BT=str(bt), FAMSEQ_ELIM_BT=str(bt),
                     FAMSEQ_LANE_MINWAVES="1", FAMSEQ_JIT_SOURCE_ONLY="1").items():
        monkeypatch.setenv(k, v)
    ctx = fs.Context(model, device=-1)
    ctx.set_option("call_kernels", 1)
    plan = ctx.plan()
    ctx.close()
    obj, entry = ((plan["elim_call_code_object"], "famseq_elim") if elim else (plan["enum_lane_call_code_object"], "famseq_enum_lane"))
    src = open(obj[:-6] + ".hip").read()
    reads_rows = "lg[" in src
    assert ("STAGE_IN_PL_FLAT" in src) == bool(elim)  # the flat staging of packed PLs: the sum-product form's
    shim = SHIM_THREADS.replace("#define famseq_enum_lane famseq_enum_lane_one_thread", "#define %s kernel_one_thread_" % entry)
    shim += SHIM_CALL + "#define __builtin_amdgcn_ballot_w64(p) ((unsigned long)(p))\n"
    src = re.sub(r"#define FS_KEEP_BRANCH\(\) asm.*", "", src).replace("typedef double fs_v2d __attribute__((ext_vector_type(2)));", "")
    src = portable(src).replace("#include <hip/hip_runtime.h>", shim)
    src = re.sub(r"#define LDS_BARRIER\(\).*", "#define LDS_BARRIER() pthread_barrier_wait(&wg_barrier_)", src)
    src = src.replace("typedef double v2d __attribute__((ext_vector_type(2)));", "").replace("__attribute__((address_space(3)))", "").replace("__attribute__((address_space(1)))", "")
    src = re.sub(r'asm volatile\(""[^\n;]*\);', "", src.replace('extern "C" __global__', "static"))
    src += """
#undef %(e)s
extern "C" void %(e)s(const double *lk, const unsigned char *fl, double *post, double *single, unsigned char *st, long n,
                      const double *tc, double lc, const fs_call_args *call) {
  pthread_barrier_init(&wg_barrier_, nullptr, BT);
  std::vector<std::thread> lanes;
  for (int t = 0; t < BT; ++t) lanes.emplace_back([=] { threadIdx.x = t; kernel_one_thread_(lk, fl, post, single, st, n, tc, lc, call); });
  for (auto &l : lanes) l.join();
  pthread_barrier_destroy(&wg_barrier_);
}
""" % dict(e=entry)
    cpp, so = str(cache / ("t%d.cpp" % elim)), str(cache / ("t%d.so" % elim))
    open(cpp, "w").write(src)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-w", "-shared", "-fPIC", "-pthread", "-o", so, cpp])
    fn = getattr(C.CDLL(so), entry)
    fn.restype = None
    fn.argtypes = [C.c_void_p] * 5 + [C.c_long, C.c_void_p, C.c_double, C.c_void_p]
    return fn, reads_rows


@pytest.mark.parametrize("seed", [2, 10, 11, 15])
def test_call_path_forms_as_whole_workgroups(seed, tmp_path, monkeypatch):
    """The call-path staging is cooperative (a lane unpacks, converts and stores other lanes' sites), so
    it is run here as a real workgroup of 8 host threads: soak pedigrees with unsequenced members, a
    shuffled column order, ONE sequenced sample (seed 11: the column division by 1), whole and ragged
    chunks, packed PLs with missing samples against fp64 rows of the table's values, and both against
    the oracle."""
    import math
    import sys

    import oracle

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _soak import soak_pedigree

    rng, ped, mu = soak_pedigree(seed)
    model = fs.make_model(ped, mrate=mu)
    n = ped.n
    seq = np.nonzero(ped.sequenced)[0].astype(np.int32)
    rng.shuffle(seq)
    k = len(seq)
    col = np.full(20, -1, np.int32)
    col[seq] = np.arange(k)
    slot = output_slots(col, n, k)
    lut = np.array([math.pow(10.0, -i / 10.0) for i in range(4096)])
    tc = np.ascontiguousarray(factor_tables(model))
    probe = fs.Context(model, device=-1)
    kinds = [False] + ([True] if probe.plan()["elim_supported"] else [])
    probe.close()
    for elim in kinds:
        fn, reads_rows = _threaded_call_kernel(model, elim, 8, tmp_path, monkeypatch)
        for S in (1, 8, 19):
            pl = rng.randint(0, 400, size=(S, k, 3)).astype(np.uint16)
            pl[np.arange(S)[:, None], np.arange(k)[None, :], rng.randint(0, 3, size=(S, k))] = 0
            pl[rng.rand(S, k) < 0.1] = 0xFFFF
            pl[rng.rand(S, k, 3) < 0.02] = 5000
            flags = rng.randint(0, 4, S).astype(np.uint8)
            table = np.append(lut, 0.0)
            lk = np.ones((S, n, 3))
            for j, mbr in enumerate(seq):
                v = table[np.minimum(pl[:, j].astype(np.int64), 4096)]
                v[np.all(pl[:, j] == 0xFFFF, axis=1)] = 1.0
                lk[:, mbr] = v

            def run(lk_in=None, pl_in=None):
                gpp, fpp = np.full((S, k, 3), -7.0), np.full((S, k, 3), -7.0)
                fgt, st = np.full((S, k), 9, np.int8), np.full(S, 77, np.uint8)
                a = CallArgs(None if pl_in is None else pl_in.ctypes.data, lut.ctypes.data, col.ctypes.data, slot.ctypes.data,
                             gpp.ctypes.data, fpp.ctypes.data, fgt.ctypes.data, k, 0xFFFFFFFF // (3 * k) + 1,
                             0xFFFFFFFF // k + 1 if k > 1 else 0)
                fn(None if lk_in is None else lk_in.ctypes.data, flags.ctypes.data, None, None, st.ctypes.data, S, tc.ctypes.data,
                   float(model.lc), C.addressof(a))
                return gpp, fpp, fgt, st

            a = run(lk_in=np.ascontiguousarray(lk))
            ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders, ped.sequenced, mrate=mu).bn_batch(lk, flags, threads=2)
            assert np.array_equal(a[3], ref[2]), (seed, elim, S)
            ok, s_ok = (ref[2] & 3) == 0, (ref[2] & 3) != 1
            np.testing.assert_allclose(a[0][s_ok], host_phred(ref[1][s_ok][:, seq]), rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(a[1][ok], host_phred(ref[0][ok][:, seq]), rtol=1e-8, atol=1e-8)
            # the call is the arg-max of the kernel's own posterior, which differs from the oracle's in the
            # last bits: where the picks differ the two candidates must be a tie to rounding
            want = fs.call_genotypes(ref[0][ok][:, seq]).reshape(-1, k)
            rows = ref[0][ok][:, seq].reshape(-1, 3)
            g_, w_ = a[2][ok].reshape(-1), want.reshape(-1)
            assert np.all(g_ >= 0)
            diff = g_ != w_
            pa, pb = rows[np.arange(len(rows)), g_], rows[np.arange(len(rows)), w_]
            assert np.all(np.abs(pa - pb)[diff] <= 1e-9 * pb[diff]), (seed, elim, S)
            assert np.all(a[2][~ok] == -1) and np.all(np.isnan(a[1][~ok]))
            if not reads_rows:
                # (16-byte aligned, as the library's device buffers are: whole chunks of the sum-product form then take the flat
                # staging — coalesced 16-byte pieces through the top of the row area, the next chunk's fetched ahead: S = 19 is two
                # whole chunks and a ragged one)
                assert pl.ctypes.data % 16 == 0
                b = run(pl_in=pl)
                for x, y in zip(a, b):
                    assert np.array_equal(x, y, equal_nan=True), (seed, elim, S)


LATE_CASES = [c for c in CASES if c.name in ("bn_synth:ped5", "bn_synth:ped5_x", "bn_synth:quad_mu0", "bn_synth:chain7", "bn_lk:fam06")]


@pytest.mark.parametrize("case", LATE_CASES, ids=[c.name for c in LATE_CASES])
def test_compute_first_lane_shell_on_wider_pedigrees(case, tmp_path, monkeypatch):
    """The lane kernel's compute-first order of phases (default for trios only, FAMSEQ_LANE_LATE=1 elsewhere):
    the enumeration's scratch slots sit behind the likelihoods in the lane's row and the marginals go through
    registers — same fixtures, looped members included."""
    monkeypatch.setenv("FAMSEQ_LANE_LATE", "1")
    model = fs.make_model(case.pedigree(), **case.consts)
    fn = build_host_kernel(model, "lane", tmp_path, monkeypatch)
    assert "compute-first shell" in open(tmp_path / "cache_lane" / "k.cpp").read()
    post, single, st = run_host(fn, model, case.lk, case.flags)
    assert np.array_equal(st, case.status)
    ok = (case.status & 3) == 0
    assert np.array_equal(single[(case.status & 3) != 1], case.single[(case.status & 3) != 1])
    np.testing.assert_allclose(post[ok], case.post[ok], rtol=1e-12, atol=0)
    assert np.all(np.isnan(post[~ok]))


@pytest.mark.parametrize("name", ["bn_synth:ped10", "bn_synth:ped10_x", "bn_vcf:fam01"])
def test_six_member_block_form_of_wide_pedigrees(name, tmp_path, monkeypatch):
    """Wide pedigrees are generated in two block sizes (kEnumVariants: a 7-member unrolled block first, the
    6-member one behind it); the host tests above see the first, the GPU picks by register spill — at ten
    members the 6-member form.  This keeps that form under the same fixtures without a GPU."""
    monkeypatch.setenv("FAMSEQ_LANE_CAP", "6")
    case = [c for c in CASES if c.name == name][0]
    model = fs.make_model(case.pedigree(), **case.consts)
    fn = build_host_kernel(model, "lane", tmp_path, monkeypatch)
    assert "6 unrolled members" in open(tmp_path / "cache_lane" / "k.cpp").readline()
    post, single, st = run_host(fn, model, case.lk, case.flags)
    assert np.array_equal(st, case.status)
    ok = (case.status & 3) == 0
    assert np.array_equal(single[(case.status & 3) != 1], case.single[(case.status & 3) != 1])
    np.testing.assert_allclose(post[ok], case.post[ok], rtol=1e-12, atol=0)
