"""include/famseq_hip.h and libfamseq_hip.so from a plain C99 program (tests/c_abi_check.c)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = str(tmp_path / "c_abi_check")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_check.c"), "-o", exe, "-L", os.path.join(ROOT, "famseq_amd", "lib"),
                           "-lfamseq_hip", "-lm", "-Wl,-rpath," + os.path.join(ROOT, "famseq_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c_program_links_and_fails_loudly_without_gpu(tmp_path):
    p = subprocess.run([build(tmp_path)], capture_output=True, text=True)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)


@pytest.mark.gpu
def test_c_program_computes_on_gpu(tmp_path):
    p = subprocess.run([build(tmp_path)], capture_output=True, text=True)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert "gpu: child posterior" in p.stdout and "22-member pedigree through famseq_create_pedigree" in p.stdout
