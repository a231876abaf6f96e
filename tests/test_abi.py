"""C-ABI surface of libfamseq_hip.so: loads without a GPU, exports every symbol the
header declares, and its host-side model setup matches the reference fixtures."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import famseq_amd as fs
import oracle
from _cases import GOLDEN, load_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "famseq_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(famseq_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = fs.lib()
    declared = header_functions()
    assert sorted(fs.ABI_SYMBOLS) == declared
    for name in declared:
        assert hasattr(L, name), name


def test_no_oracle_in_product():
    """The product must not link or reference the oracle."""
    so = open(fs.LIB_PATH, "rb").read()
    assert b"oracle_bn" not in so and b"liboracle" not in so
    for root, _, files in os.walk(os.path.join(ROOT, "famseq_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "bn_oracle" not in txt, f


def test_tables_bit_exact_vs_reference():
    z = np.load(os.path.join(GOLDEN, "tables.npz"))
    for key in [k for k in z.files if k.endswith(".pcp2")]:
        mu = float(key[2:-5])
        a, b, c = fs.transmission_tables(mu)
        for got, ref in ((a, z[key]), (b, z[key[:-5] + ".xf"]), (c, z[key[:-5] + ".xm"])):
            assert np.array_equal(got.view(np.uint64), ref.view(np.uint64)), (mu, key)


@pytest.mark.parametrize("c", load_cases(("bn_vcf.npz", "bn_synth.npz")), ids=repr)
def test_model_init_matches_oracle(c):
    m = fs.make_model(c.pedigree(), **c.consts)
    o = oracle.OracleModel(c.ids, c.mids, c.fids, c.genders, c.sequenced, **c.consts)
    n = c.n
    assert m.n_members == n
    assert list(m.mother[:n]) == o.mother.tolist() and list(m.father[:n]) == o.father.tolist()
    assert list(m.sequenced[:n]) == list(c.sequenced)
    for name in ("pcp2", "pcp2Xf", "pcp2Xm", "genoProbN", "genoProbK", "genoProbXN", "genoProbXK"):
        assert np.array_equal(np.array(getattr(m, name)[:]), o.table(name)), name
    assert m.lc == o.c.lc


def test_model_init_rejects_like_the_reference():
    half = fs.Pedigree([1, 2, 3], [0, 0, 2], [0, 0, 0], [1, 2, 1], ["a", "b", "c"])
    with pytest.raises(fs.FamseqError, match="fulfill"):
        fs.make_model(half)
    sex = fs.Pedigree([1, 2, 3], [0, 0, 1], [0, 0, 2], [1, 2, 1], ["a", "b", "c"])  # mother is male
    with pytest.raises(fs.FamseqError):
        fs.make_model(sex)
    with pytest.raises(ValueError):
        half.relations()
    with pytest.raises(ValueError):
        sex.relations()


def test_create_rejects_a_pedigree_with_a_cycle():
    """Not a case the reference guards (it would walk such a file as it stands); the generators
    order members by generation, so a member who is their own ancestor is refused up front."""
    ped = fs.Pedigree([1, 2, 3, 4], [3, 0, 1, 0], [4, 0, 2, 0], [2, 1, 2, 1], ["a", "b", "c", "d"])
    with pytest.raises(fs.FamseqError, match="own ancestor"):
        fs.Context(fs.make_model(ped), device=-1)


def test_call_genotypes_first_max_wins():
    g = fs.call_genotypes([[0.2, 0.5, 0.3], [0.5, 0.5, 0.0], [0.1, 0.1, 0.8], [float("nan")] * 3])
    assert g.tolist() == [1, 0, 2, -1]


def test_plan_only_context_refuses_to_compute():
    m = fs.make_model(fs.synthetic_pedigree("ped5"))
    ctx = fs.Context(m, device=-1)
    lk = np.ones((4, 5, 3))
    with pytest.raises(fs.FamseqError, match="no CPU path"):
        ctx.bn_batch(lk)
    ctx.close()


@pytest.mark.skipif(fs.device_count() > 0, reason="only meaningful without a GPU")
def test_fails_loudly_without_gpu():
    m = fs.make_model(fs.synthetic_pedigree("ped5"))
    with pytest.raises(fs.FamseqError, match="no usable HIP device"):
        fs.Context(m, device=0)


def test_host_code_is_clean_under_asan_and_ubsan():
    """`make asan` (SURVEY.md section 5's sanitizer recipe): model, plan, both generators, the JIT and the C
    ABI rebuilt under AddressSanitizer + UBSan and driven on plan-only contexts over eight pedigrees of
    1-20 members, the rejections and the no-compiler path.  CPU only."""
    import subprocess

    r = subprocess.run(["make", "-s", "-C", ROOT, "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "asan_host_check: ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def _compile_in_subprocess(tmp_path, extra_env):
    """A plan-only context compiles the trio's sum-product kernel into an empty cache; (returncode, output, files)."""
    import subprocess
    import sys

    code = ("import famseq_amd as fs\n"
            "ctx = fs.Context(fs.make_model(fs.synthetic_pedigree('trio')), device=-1)\n"
            "ctx.set_option('engine', fs.ENGINE_ELIM)\n"
            "print(ctx.plan()['elim_code_object'])\n")
    env = dict(os.environ, FAMSEQ_KERNEL_CACHE=str(tmp_path), FAMSEQ_QUIET="1", **extra_env)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    return p.returncode, p.stdout + p.stderr, sorted(f.name for f in tmp_path.iterdir())


def test_kernels_compile_in_process_without_hipcc(tmp_path):
    """jit.cpp compiles through libhiprtc (part of the HIP runtime): no hipcc on the host, no child process (so also
    where a profiler is attached and a process that has initialised the GPU must not exec: tests/test_gpu_multi.py)."""
    rc, out, files = _compile_in_subprocess(tmp_path, {"FAMSEQ_HIPCC": "/nonexistent/hipcc"})
    assert rc == 0, out
    assert any(f.endswith(".hsaco") for f in files) and all(f.endswith((".hsaco", ".res")) for f in files), files
    obj = [f for f in files if f.endswith(".hsaco")][0]
    assert open(os.path.join(str(tmp_path), obj), "rb").read(4) == b"\x7fELF"


def test_no_compiler_at_all_is_an_error_not_a_silent_path(tmp_path):
    rc, out, files = _compile_in_subprocess(tmp_path, {"FAMSEQ_HIPCC": "/nonexistent/hipcc", "FAMSEQ_NO_HIPRTC": "1"})
    assert rc != 0 and "compilation failed" in out, out
    assert not any(f.endswith(".hsaco") for f in files), files


def test_tune_needs_a_device():
    """famseq_set_option "tune" times kernels on the GPU: a plan-only context says so (and there is no CPU path to time)."""
    ctx = fs.Context(fs.make_model(fs.synthetic_pedigree("trio")), device=-1)
    with pytest.raises(fs.FamseqError, match="needs a device"):
        ctx.set_option("tune", 1)
    ctx.close()
