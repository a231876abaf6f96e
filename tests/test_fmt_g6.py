"""The CLI's number formatter (csrc/host/fmt_g6.h) prints exactly what printf("%g") prints — i.e. what the reference's
`ostream << double` writes for every GPP / FPP value (file.cpp:702-731): exact ties, one ulp either side of powers of ten
and of six-digit decimals, the range edges, and 1.5 M random Phred-shaped values (tests/fmt_g6_check.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_g6_equals_printf(tmp_path):
    exe = str(tmp_path / "fmt_g6_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "famseq_amd", "csrc", "host"),
                           os.path.join(ROOT, "tests", "fmt_g6_check.cpp"), "-o", exe])
    p = subprocess.run([exe, "500000"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:]
    assert " 0 mismatches" in p.stdout


def test_g6_under_sanitizers(tmp_path):
    """The same check under AddressSanitizer + UBSan.  Round 2's formatter read its power-of-ten table at index -1 for
    values in [2^19, 1e6) and was right by the luck of what lay in front of the table; the shared core (csrc/g6_core.h)
    caps the exponent instead."""
    exe = str(tmp_path / "fmt_g6_check_asan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(ROOT, "famseq_amd", "csrc", "host"), os.path.join(ROOT, "tests", "fmt_g6_check.cpp"), "-o", exe])
    p = subprocess.run([exe, "50000"], capture_output=True, text=True)
    assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
    assert " 0 mismatches" in p.stdout
