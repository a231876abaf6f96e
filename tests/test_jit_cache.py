"""The kernel cache's corner cases (no GPU: plan-only contexts compile into a scratch cache):
a failed call-path build leaves the plain kernels alone, placeholders and empty files are no cache hits,
a local tuning note overrides the shipped one, and a measured pick is loaded exactly."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

import famseq_amd as fs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_py(code, env_extra, tmp_path):
    env = dict(os.environ, FAMSEQ_NO_TORCH="1", FAMSEQ_QUIET="1", PYTHONPATH=ROOT)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_a_failed_call_path_build_does_not_disable_the_plain_kernels(tmp_path):
    cache = tmp_path / "cache"
    cache.mkdir()
    warm = "import famseq_amd as fs\nc = fs.Context(fs.make_model(fs.synthetic_pedigree('quad')), device=-1)\n" \
           "c.set_option('enum_impl', 1); c.set_option('engine', fs.ENGINE_ELIM); print(c.plan()['enum_lane_code_object'])\n"
    run_py(warm, {"FAMSEQ_KERNEL_CACHE": str(cache)}, tmp_path)  # plain lane + sum-product kernels are in the cache now
    code = r"""
import famseq_amd as fs
c = fs.Context(fs.make_model(fs.synthetic_pedigree('quad')), device=-1)
try:
    c.set_option('call_kernels', 1)
    raise SystemExit('the call-path forms built without a compiler?')
except fs.FamseqError as e:
    assert 'call-path kernel unavailable' in str(e), e
c.set_option('enum_impl', 1)              # ... and the plain kernels still load from the cache
c.set_option('engine', fs.ENGINE_ELIM)
p = c.plan()
assert p['enum_lane_failed'] == 0 and p['enum_lane_code_object'] and p['elim_code_object'], p
assert p['enum_lane_call_error'] and not p['enum_lane_call_code_object'], p
print('ok')
"""
    out = run_py(code, {"FAMSEQ_KERNEL_CACHE": str(cache), "FAMSEQ_NO_HIPRTC": "1", "FAMSEQ_HIPCC": "/nonexistent/hipcc"}, tmp_path)
    assert "ok" in out


def test_placeholders_and_empty_files_are_not_cache_hits(tmp_path):
    cache = tmp_path / "cache"
    cache.mkdir()
    code = "import famseq_amd as fs\nc = fs.Context(fs.make_model(fs.synthetic_pedigree('trio')), device=-1)\n" \
           "c.set_option('enum_impl', 1); print(c.plan()['enum_lane_code_object'])\n"
    obj = run_py(code, {"FAMSEQ_KERNEL_CACHE": str(cache), "FAMSEQ_JIT_SOURCE_ONLY": "1", "FAMSEQ_KEEP_SRC": "1"}, tmp_path).strip()
    assert obj.endswith(".hsaco") and not os.path.exists(obj) and os.path.exists(obj[:-6] + ".hip")  # source kept, no object
    open(obj, "wb").close()  # an empty object (an interrupted writer): must be compiled over, not taken
    obj2 = run_py(code, {"FAMSEQ_KERNEL_CACHE": str(cache)}, tmp_path).strip()
    assert obj2 == obj and os.path.getsize(obj) > 1000
    # without an explicit cache directory the source-only switch is ignored (it cannot poison a shared cache)
    before = set(glob.glob(os.path.join(ROOT, "famseq_amd", "lib", "kernels", "*.hsaco")))
    obj3 = run_py(code, {"FAMSEQ_JIT_SOURCE_ONLY": "1"}, tmp_path).strip()
    assert os.path.getsize(obj3) > 1000
    assert all(os.path.getsize(f) > 0 for f in set(glob.glob(os.path.join(ROOT, "famseq_amd", "lib", "kernels", "*.hsaco"))) - before)


def test_a_measured_pick_is_loaded_exactly(tmp_path):
    cache = tmp_path / "cache"
    cache.mkdir()
    code = r"""
import famseq_amd as fs
m = fs.make_model(fs.synthetic_pedigree('ped10'))
for lane, elim in ((2, 1), (0, 0), (3, 2)):
    c = fs.Context(m, device=-1)
    c.set_option('pick_lane', lane); c.set_option('pick_elim', elim)
    c.set_option('enum_impl', 1); c.set_option('engine', fs.ENGINE_ELIM)
    p = c.plan()
    assert (p['enum_lane_variant'], p['elim_variant']) == (lane, elim), p
    c.close()
print('ok')
"""
    assert "ok" in run_py(code, {"FAMSEQ_KERNEL_CACHE": str(cache)}, tmp_path)


def test_a_local_tuning_note_overrides_the_shipped_one(tmp_path):
    """A deployment whose library directory is read-only and ships a pick: the user's own note lands in the per-user
    cache and is the one read back."""
    lib = tmp_path / "lib"
    (lib / "kernels").mkdir(parents=True)
    shutil.copy(fs.LIB_PATH, lib / "libfamseq_hip.so")
    code = r"""
import ctypes as C, os, sys
import famseq_amd as fs
fs.LIB_PATH = sys.argv[1]
m = fs.make_model(fs.synthetic_pedigree('quad'))
c = fs.Context(m, device=-1)
c.set_option(sys.argv[2], int(sys.argv[3]))
c.set_option('engine', fs.ENGINE_ELIM)
print(c.plan()['elim_variant'])
"""
    script = tmp_path / "t.py"
    script.write_text(code)
    env = dict(os.environ, FAMSEQ_NO_TORCH="1", PYTHONPATH=ROOT)
    env.pop("FAMSEQ_KERNEL_CACHE", None)

    def run(*args):
        r = subprocess.run([sys.executable, str(script), str(lib / "libfamseq_hip.so")] + list(args), env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout.strip().splitlines()[-1]

    assert run("pick_elim", "1") == "1"          # written into lib/kernels (still writable): the "shipped" pick
    os.chmod(lib / "kernels", 0o555)
    env["FAMSEQ_LIBDIR_READONLY"] = "1"           # (root ignores the mode bits: say it in words as well)
    try:
        assert run("pick_elim", "0") == "0"      # lands in /tmp/famseq_kernels_<uid> and wins over the shipped 1
        # ... and a context that sets nothing reads the local note, not the shipped one
        script.write_text(code.replace("c.set_option(sys.argv[2], int(sys.argv[3]))", "pass"))
        assert run("-", "0") == "0"
    finally:
        os.chmod(lib / "kernels", 0o755)
        for f in glob.glob("/tmp/famseq_kernels_%d/*" % os.getuid()):
            os.unlink(f)
