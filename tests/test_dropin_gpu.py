"""The link-time drop-in, run for real: oracle/_ref/FamSeq_hipref is the
REFERENCE's own FamSeq.cpp / file.cpp / checkInput.cpp / normal.cpp / family.cpp, compiled where
they lie, with the single symbol family::calPostProbBN(bool,int) (family.h:375) taken from
oracle/family_hip.cpp, which forwards it to libfamseq_hip.so through the C ABI — the substitution
the reference itself makes between makefile:4 and makefile.gpu:83-87.  The reference's drivers
(callGenoMVCF file.cpp:595->607->680, callGenoLK :1743->1751->1804) then run on the GPU path, one
site per call, and their output is compared with the reference CPU CLI's (tests/golden/ref_cli).
The binary is test infrastructure (built by `make -C oracle ref` in the build container; it travels
to the GPU box as a binary, like oracle/_ref/libfamseq_ref.so)."""
import os
import subprocess

import pytest

from test_cli_gpu import REF, TD, assert_same_output

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPREF = os.path.join(ROOT, "oracle", "_ref", "FamSeq_hipref")


def run_hipref(args, out):
    if not os.path.exists(HIPREF):
        pytest.skip("oracle/_ref/FamSeq_hipref was not built (needs /root/reference at build time)")
    p = subprocess.run([HIPREF] + args + ["-output", str(out)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "family_hip:" not in p.stderr, p.stderr[-2000:]
    return p.stdout


@pytest.mark.parametrize("fam", range(1, 7))
@pytest.mark.parametrize("tag,extra", [("v", ["-v"]), ("a", ["-a"]), ("plain", [])])
def test_reference_vcf_driver_on_the_hip_operator(fam, tag, extra, tmp_path):
    out = tmp_path / "o.vcf"
    run_hipref(["vcf", "-vcfFile", TD + "/test_subset.vcf", "-pedFile", "%s/fam%02d.ped" % (TD, fam), "-method", "1"] + extra, out)
    assert assert_same_output(out, "%s/subset_fam%02d_%s.vcf" % (REF, fam, tag)) >= 12


@pytest.mark.parametrize("fam", range(1, 7))
def test_reference_lk_driver_on_the_hip_operator(fam, tmp_path):
    out = tmp_path / "o.txt"
    run_hipref(["LK", "-lkFile", TD + "/loftest.txt", "-pedFile", "%s/fam%02d.ped" % (TD, fam), "-method", "1"], out)
    assert assert_same_output(out, "%s/loftest_fam%02d.txt" % (REF, fam)) == 100


@pytest.mark.parametrize("tag,extra", [("default", []), ("mu0", ["-mRate", "0", "-a"]),
                                       ("priors", ["-genoProbN", "0.9", "0.08", "0.02", "-genoProbK", "0.3", "0.4", "0.3",
                                                   "-genoProbXN", "0.97", "0.03", "-genoProbXK", "0.6", "0.4", "-mRate",
                                                   "1e-3", "-LRC", "0.9"])])
def test_reference_driver_probes(tag, extra, tmp_path):
    """chrX, Known, both failure kinds (the `false` return and its :NA:NA:NA lines), missing sample,
    GL, custom priors / mutation rate / -LRC — all through the reference's driver."""
    out = tmp_path / "o.vcf"
    stdout = run_hipref(["vcf", "-vcfFile", TD + "/probe.vcf", "-pedFile", TD + "/probe.ped", "-method", "1"] + extra, out)
    assert_same_output(out, "%s/probe_%s.vcf" % (REF, tag))
    ref_na = open("%s/probe_%s.vcf" % (REF, tag)).read().count(":NA:NA:NA\t") // 4
    assert stdout.count("Warning: this variant hasn't been calculated") == ref_na
