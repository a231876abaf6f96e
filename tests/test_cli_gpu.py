"""End-to-end text parity of bin/FamSeq (our CLI over the C ABI) with outputs of the compiled
reference CLI (tests/golden/ref_cli, made by oracle/gen_golden.py).

Comparison rule: header and pass-through lines byte-identical; in result
lines every field identical except the Phred numbers, which the reference prints with 6
significant digits: |a-b| <= 1e-5*max(|a|,|b|) or, for values < 1e-3 (the -10*log10(1-eps)
rounding noise), <= 1e-6 absolute; `99999`, `NA` and the FGT call must match exactly."""
import os
import subprocess

import pytest

from famseq_amd import plfile

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bin", "FamSeq")
TD = os.path.join(ROOT, "tests", "golden", "testdata")
REF = os.path.join(ROOT, "tests", "golden", "ref_cli")


def run_cli(args, out):
    p = subprocess.run([CLI] + args + ["-output", str(out)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def num_close(a, b):
    if a == b:
        return True
    if "99999" in (a, b) or "NA" in (a, b):
        return False
    x, y = float(a), float(b)
    if max(abs(x), abs(y)) < 1e-3:
        return abs(x - y) <= 1e-6
    return abs(x - y) <= 1e-5 * max(abs(x), abs(y))


def assert_same_output(got_path, ref_path, tags=":GPP:FPP:FGT"):
    got = open(got_path).read().split("\n")
    ref = open(ref_path).read().split("\n")
    assert len(got) == len(ref), "line count %d vs %d" % (len(got), len(ref))
    n_results = 0
    for ln, (g, r) in enumerate(zip(got, ref), 1):
        if g == r:
            n_results += tags in r or r.startswith("LK:GPP")
            continue
        gt, rt = g.split("\t"), r.split("\t")
        assert len(gt) == len(rt), "line %d: column count" % ln
        for a, b in zip(gt, rt):
            if a == b:
                continue
            ga, rb = a.split(":"), b.split(":")
            assert len(ga) == len(rb), "line %d: %r vs %r" % (ln, a, b)
            for u, v in zip(ga, rb):
                if u == v:
                    continue
                us, vs = u.split(","), v.split(",")
                assert len(us) == len(vs) == 3, "line %d: %r vs %r" % (ln, a, b)
                assert all(num_close(p, q) for p, q in zip(us, vs)), "line %d: %r vs %r" % (ln, a, b)
        n_results += 1
    return n_results


@pytest.mark.parametrize("fam", range(1, 7))
@pytest.mark.parametrize("tag,extra", [("v", ["-v"]), ("a", ["-a"]), ("plain", [])])
def test_testdata_vcf(fam, tag, extra, tmp_path):
    """TestData pedigrees x the reduced test.vcf (every PL-bearing line of TestData/test.vcf)."""
    out = tmp_path / "o.vcf"
    run_cli(["vcf", "-vcfFile", TD + "/test_subset.vcf", "-pedFile", "%s/fam%02d.ped" % (TD, fam), "-method", "1"] + extra, out)
    assert assert_same_output(out, "%s/subset_fam%02d_%s.vcf" % (REF, fam, tag)) >= 12


def test_baseline_config_1_result_lines(tmp_path):
    """BASELINE config #1 (TestData/test.vcf + fam01.ped, -method 1 -v): the reference's 124 header
    + 12 result lines, reproduced from the reduced VCF (which holds every computable line)."""
    out = tmp_path / "o.vcf"
    run_cli(["vcf", "-vcfFile", TD + "/test_subset.vcf", "-pedFile", TD + "/fam01.ped", "-v"], out)
    assert assert_same_output(out, REF + "/full_fam01_v.vcf") == 12


@pytest.mark.parametrize("tag,extra,n_lines", [("plain", [], 9749), ("a", ["-a"], 10008)])
@pytest.mark.parametrize("batch", [None, "7", "1000"])
def test_baseline_config_1_full_length(tag, extra, n_lines, batch, tmp_path):
    """BASELINE config #1 over the WHOLE TestData/test.vcf (kept gzipped as a fixture), without -v:
    the header echo with its one-line lag, the echo / drop / all-missing / no-PL rules
    (file.cpp:143-196, :362-555) and the order-preserving block writer over ~10 k lines.  Line counts
    are SURVEY.md App. E's.  With FAMSEQ_BATCH=7 (parser blocks and GPU flushes of 7 lines) echo
    lines and sites straddle hundreds of block boundaries; the default keeps it in one block."""
    import gzip
    import shutil

    vcf, ref, out = tmp_path / "test.vcf", tmp_path / "ref.vcf", tmp_path / "o.vcf"
    for src, dst in ((TD + "/test_full.vcf.gz", vcf), ("%s/full_fam01_%s.vcf.gz" % (REF, tag), ref)):
        with gzip.open(src, "rb") as f, open(dst, "wb") as g:
            shutil.copyfileobj(f, g)
    env = dict(os.environ) if batch is None else dict(os.environ, FAMSEQ_BATCH=batch)
    p = subprocess.run([CLI, "vcf", "-vcfFile", str(vcf), "-pedFile", TD + "/fam01.ped", "-method", "1", "-output", str(out)] + extra,
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert sum(1 for _ in open(ref)) == n_lines
    assert assert_same_output(out, ref) == 12


@pytest.mark.parametrize("fam", range(1, 7))
def test_testdata_lk(fam, tmp_path):
    out = tmp_path / "o.txt"
    run_cli(["LK", "-lkFile", TD + "/loftest.txt", "-pedFile", "%s/fam%02d.ped" % (TD, fam)], out)
    assert assert_same_output(out, "%s/loftest_fam%02d.txt" % (REF, fam)) == 100


@pytest.mark.parametrize("typ", ["log10", "ln", "PS"])
def test_lk_types(typ, tmp_path):
    out = tmp_path / "o.txt"
    run_cli(["LK", "-lkFile", "%s/lk_%s.txt" % (TD, typ), "-pedFile", TD + "/fam04.ped", "-lkType", typ], out)
    assert assert_same_output(out, "%s/lk_%s_fam04.txt" % (REF, typ)) == 20


PROBES = {
    "default": [], "a": ["-a"], "v": ["-v"], "mu0": ["-mRate", "0", "-a"],
    "priors": ["-genoProbN", "0.9", "0.08", "0.02", "-genoProbK", "0.3", "0.4", "0.3", "-genoProbXN", "0.97", "0.03",
               "-genoProbXK", "0.6", "0.4", "-mRate", "1e-3", "-LRC", "0.9"],
    "loc": ["-l", TD + "/probe.loc"],
}


@pytest.mark.parametrize("tag", sorted(PROBES))
def test_probe_vcf(tag, tmp_path):
    """chrX / Known / status 1 / status 2 (-mRate 0) / missing sample / GL / echo and drop rules /
    location filter / custom priors, against the reference CLI's output."""
    out = tmp_path / "o.vcf"
    stdout = run_cli(["vcf", "-vcfFile", TD + "/probe.vcf", "-pedFile", TD + "/probe.ped"] + PROBES[tag], out)
    assert_same_output(out, "%s/probe_%s.vcf" % (REF, tag))
    ref_na = open("%s/probe_%s.vcf" % (REF, tag)).read().count(":NA:NA:NA\t") // 4
    assert stdout.count("Warning: this variant hasn't been calculated") == ref_na
    if tag == "mu0":
        assert ref_na == 2  # POS 400 (single posterior fails) and POS 500 (all 81 terms are 0)


def test_header_fallback_lines(tmp_path):
    out = tmp_path / "o.vcf"
    run_cli(["vcf", "-vcfFile", TD + "/probe_noanchor.vcf", "-pedFile", TD + "/probe.ped"], out)
    assert_same_output(out, REF + "/probe_noanchor.vcf")
    assert "calbulated by Single Method" in open(out).read()


def test_small_batches_keep_output_order(tmp_path):
    out = tmp_path / "o.vcf"
    env = dict(os.environ, FAMSEQ_BATCH="3")
    p = subprocess.run([CLI, "vcf", "-vcfFile", TD + "/test_subset.vcf", "-pedFile", TD + "/fam06.ped", "-a", "-output", str(out)],
                       capture_output=True, text=True, env=env)
    assert p.returncode == 0
    assert_same_output(out, REF + "/subset_fam06_a.vcf")


@pytest.mark.parametrize("tag,vcf,ped", [("probe", "probe.vcf", "probe.ped"), ("subset_fam01", "test_subset.vcf", "fam01.ped"),
                                         ("subset_fam04", "test_subset.vcf", "fam04.ped")])
def test_method_2_matches_reference_peeling(tag, vcf, ped, tmp_path):
    """-method 2 (Elston-Stewart in the reference) is served by the exact sum-product engine."""
    out = tmp_path / "o.vcf"
    run_cli(["vcf", "-vcfFile", TD + "/" + vcf, "-pedFile", TD + "/" + ped, "-method", "2", "-a"], out)
    assert_same_output(out, "%s/%s_method2.vcf" % (REF, tag))
    out = tmp_path / "o.txt"
    run_cli(["LK", "-lkFile", TD + "/loftest.txt", "-pedFile", TD + "/fam01.ped", "-method", "2"], out)
    assert assert_same_output(out, REF + "/loftest_fam01_method2.txt") == 100


def test_usage_and_errors(tmp_path):
    assert subprocess.run([CLI]).returncode == 255
    assert subprocess.run([CLI, "bogus"], capture_output=True).returncode == 255
    p = subprocess.run([CLI, "vcf", "-vcfFile", TD + "/probe.vcf", "-output", str(tmp_path / "x")], capture_output=True, text=True)
    assert p.returncode == 255 and "ped file must be set" in p.stdout
    p = subprocess.run([CLI, "vcf", "-vcfFile", TD + "/probe.vcf", "-pedFile", TD + "/probe.ped", "-output", str(tmp_path / "x"),
                        "-method", "3"], capture_output=True, text=True)
    assert p.returncode == 255 and "-method 3" in p.stdout


def result_fields(path, prefix):
    """[(GPP, FPP, FGT) per sample] per result line."""
    out = []
    for line in open(path):
        if line.startswith("#") or ":GPP:FPP:FGT" not in line:
            continue
        t = line.rstrip("\n").split("\t")
        samples = [x for x in (t[9:] if prefix == "vcf" else t[1:]) if x]
        out.append([tuple(x.split(":")[-3:]) for x in samples])
    return out


@pytest.mark.parametrize("vcf,ped", [("test_subset.vcf", "fam01.ped"), ("test_subset.vcf", "fam04.ped"), ("probe.vcf", "probe.ped")])
def test_packed_pl_pipeline_gives_the_vcf_results(vcf, ped, tmp_path):
    """vcf -> `FamSeq pack` -> .fspl -> `FamSeq PL`: same GPP/FPP/FGT as `FamSeq vcf` on the sites
    that are packable (integer PLs)."""
    v, p, b, o = tmp_path / "v.vcf", tmp_path / "p.txt", tmp_path / "x.fspl", TD + "/" + ped
    run_cli(["vcf", "-vcfFile", TD + "/" + vcf, "-pedFile", o], v)
    r = subprocess.run([CLI, "pack", "-vcfFile", TD + "/" + vcf, "-pedFile", o, "-output", str(b)], capture_output=True, text=True)
    assert r.returncode == 0
    run_cli(["PL", "-plFile", str(b), "-pedFile", o], p)
    a = [x for x, line in zip(result_fields(v, "vcf"), [l for l in open(v) if ":GPP:FPP:FGT" in l and not l.startswith("#")])
         if "GT:GL" not in line]
    got = result_fields(p, "pl")
    assert len(got) == len(a) > 0
    for x, y in zip(a, got):
        assert len(x) == len(y)
        for (g1, f1, t1), (g2, f2, t2) in zip(x, y):
            assert t1 == t2
            for u, w in zip(g1.split(",") + f1.split(","), g2.split(",") + f2.split(",")):
                assert num_close(u, w), (x, y)


@pytest.mark.parametrize("vcf,ped", [("test_subset.vcf", "fam01.ped"), ("probe.vcf", "probe.ped")])
def test_binary_result_file_unpacks_to_the_same_text(vcf, ped, tmp_path):
    """`FamSeq PL -binOutput` writes what the GPU hands back; `FamSeq unpack` turns it into the very
    text `FamSeq PL` prints (byte for byte), without a GPU context.  Small batches: several blocks."""
    b, txt, po, back, o = tmp_path / "x.fspl", tmp_path / "p.txt", tmp_path / "r.fspo", tmp_path / "u.txt", TD + "/" + ped
    r = subprocess.run([CLI, "pack", "-vcfFile", TD + "/" + vcf, "-pedFile", o, "-output", str(b)], capture_output=True, text=True)
    assert r.returncode == 0
    env = dict(os.environ, FAMSEQ_BATCH="5")
    for args, out in ((["PL", "-plFile", str(b), "-pedFile", o], txt),
                      (["PL", "-plFile", str(b), "-pedFile", o, "-binOutput"], po),
                      (["unpack", "-plFile", str(b), "-poFile", str(po), "-pedFile", o], back)):
        p = subprocess.run([CLI] + args + ["-output", str(out)], capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
    assert open(po, "rb").read(8) == b"FSPO0001"
    assert open(back).read() == open(txt).read() and os.path.getsize(txt) > 500
    # and the Python reader sees the same numbers
    res = plfile.read_results(str(po))
    assert res["status"].shape[0] == sum(1 for line in open(txt) if not line.startswith("#"))


def test_packed_pipeline_edge_sizes(tmp_path):
    """`FamSeq PL` (reader / GPU / writer threads over three rotating batches): an empty packed file, fewer
    sites than one batch, exactly three batches and a ragged fourth — text and packed output alike; the packed
    results unpack to the text."""
    import numpy as np

    from famseq_amd import pedigree, synth

    ped = pedigree.synthetic_pedigree("ped5")
    mo, fa = ped.relations()
    pedf = tmp_path / "p.ped"
    pedigree.write_ped(ped, str(pedf))
    for n, batch in ((0, "4"), (3, "4"), (12, "4"), (14, "4"), (700, "256")):
        pl, known, _ = synth.gen_sites(mo, fa, max(n, 1), synth.SEED_BASE + 1)
        fspl, txt, po, back = tmp_path / ("s%d.fspl" % n), tmp_path / ("t%d.txt" % n), tmp_path / ("r%d.fspo" % n), tmp_path / ("u%d.txt" % n)
        plfile.write_plfile(str(fspl), ped.names, known[:n].astype(np.uint8), pl[:n].astype(np.uint16))
        env = dict(os.environ, FAMSEQ_BATCH=batch)
        for args, out in ((["PL", "-plFile", str(fspl), "-pedFile", str(pedf)], txt),
                          (["PL", "-plFile", str(fspl), "-pedFile", str(pedf), "-binOutput"], po),
                          (["unpack", "-plFile", str(fspl), "-poFile", str(po), "-pedFile", str(pedf)], back)):
            p = subprocess.run([CLI] + args + ["-output", str(out)], capture_output=True, text=True, timeout=300, env=env)
            assert p.returncode == 0, (n, args, p.stdout + p.stderr)
        lines = [l for l in open(txt) if not l.startswith("#")]
        assert len(lines) == n
        assert open(back).read() == open(txt).read()
        res = plfile.read_results(str(po))
        assert res["status"].shape[0] == n


def test_blocks_parsed_on_many_threads_give_the_single_thread_text(tmp_path):
    """The vcf driver parses and formats a block of >= 2048 lines on all cores, each thread its own range; a block
    with GL (non-integer) fields goes to the GPU as fp64 rows rebuilt from every thread's packed integers plus the
    parsed rows.  probe.vcf's body 200 times over (4,600 lines: every echo / drop / failure / chrX / GL rule in
    every thread's range) must print the single run's body 200 times over, warnings included."""
    lines = open(TD + "/probe.vcf").read().split("\n")
    head = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if l and not l.startswith("#")]
    big = tmp_path / "big.vcf"
    big.write_text("\n".join(head + body * 200) + "\n")
    one, many, serial = tmp_path / "one.vcf", tmp_path / "many.vcf", tmp_path / "serial.vcf"
    args = ["-pedFile", TD + "/probe.ped", "-a", "-mRate", "0"]
    w1 = run_cli(["vcf", "-vcfFile", TD + "/probe.vcf"] + args, one)
    wn = run_cli(["vcf", "-vcfFile", str(big)] + args, many)
    p = subprocess.run([CLI, "vcf", "-vcfFile", str(big)] + args + ["-output", str(serial)], capture_output=True, text=True,
                       env=dict(os.environ, FAMSEQ_THREADS="1", FAMSEQ_BATCH="1000"))
    assert p.returncode == 0
    # (compared as everywhere in this file: a 23-site batch and a 4,600-site batch run different kernels, whose
    # posteriors differ in the last bits, which -10 log10(1 - 1e-12) magnifies into the fourth digit of a 4e-12)
    text_one = open(one).read().split("\n")
    head_one = [l for l in text_one if l.startswith("#")]
    body_one = [l for l in text_one if l and not l.startswith("#")]
    expect = tmp_path / "expect.vcf"
    expect.write_text("\n".join(head_one + body_one * 200) + "\n")
    assert assert_same_output(many, expect) == 200 * assert_same_output(one, one) > 0
    assert_same_output(serial, expect)
    n_warn = w1.count("Warning: this variant hasn't been calculated")
    assert n_warn == 2 and wn.count("Warning: this variant hasn't been calculated") == 200 * n_warn


@pytest.mark.parametrize("batch", [None, "5"])
def test_sites_moved_together_for_large_pedigrees(batch, tmp_path):
    """From twelve members on the vcf driver compacts a block's sites before the GPU call (a line that is no site
    would cost 3^N configurations as a dummy).  No TestData pedigree is that large: FAMSEQ_COMPACT_FROM=1 turns the
    path on for the probe file (echo / drop / failure / chrX / missing-sample / GL lines between the sites, the GL
    line taking the block through the fp64 rows) and for TestData's full VCF (12 sites among 9,861 lines)."""
    env = dict(os.environ, FAMSEQ_COMPACT_FROM="1")
    if batch:
        env["FAMSEQ_BATCH"] = batch
    for args, ref in ((["-vcfFile", TD + "/probe.vcf", "-pedFile", TD + "/probe.ped", "-a", "-mRate", "0"], REF + "/probe_mu0.vcf"),
                      (["-vcfFile", TD + "/probe.vcf", "-pedFile", TD + "/probe.ped"], REF + "/probe_default.vcf"),
                      (["-vcfFile", TD + "/test_subset.vcf", "-pedFile", TD + "/fam01.ped", "-a"], REF + "/subset_fam01_a.vcf")):
        out = tmp_path / "o.vcf"
        p = subprocess.run([CLI, "vcf"] + args + ["-output", str(out)], capture_output=True, text=True, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        assert assert_same_output(out, ref) > 0


@pytest.mark.parametrize("vcf,ped,extra,batch", [("test_full.vcf.gz", "fam01.ped", ["-a"], None), ("test_full.vcf.gz", "fam04.ped", [], "1000"),
                                                 ("probe.vcf", "probe.ped", ["-a"], "3"), ("test_subset.vcf", "fam06.ped", ["-v"], None)])
def test_device_text_is_the_host_formatter_text(vcf, ped, extra, batch, tmp_path):
    """The numbers are turned into text on the device (famseq_bn_call_text_batch, the default) or by the host formatter
    (FAMSEQ_HOST_FORMAT=1: csrc/host/fmt_g6.h, itself checked against printf): the same bytes, failed sites, missing samples
    and echoed lines included."""
    import gzip
    import shutil

    src = TD + "/" + vcf
    if vcf.endswith(".gz"):
        src = str(tmp_path / "in.vcf")
        with gzip.open(TD + "/" + vcf, "rb") as f, open(src, "wb") as g:
            shutil.copyfileobj(f, g)
    outs = []
    for host in ("0", "1"):
        env = dict(os.environ, FAMSEQ_HOST_FORMAT=host)
        if batch:
            env["FAMSEQ_BATCH"] = batch
        out = tmp_path / ("o%s.vcf" % host)
        p = subprocess.run([CLI, "vcf", "-vcfFile", src, "-pedFile", TD + "/" + ped, "-output", str(out)] + extra, capture_output=True,
                           text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and outs[0].count(b":GPP:FPP:FGT") >= 10


@pytest.mark.parametrize("first", [0, 8, 16])
def test_differential_fuzz_against_the_compiled_reference_cli(first):
    """tools/cli_fuzz.py, eight seeds per case: random pedigrees (loops, unsequenced members, a sample that is not in the PED),
    random VCF lines of every kind the reference's driver tells apart, random -v / -a / -LRC / -mRate — `bin/FamSeq vcf` against
    the reference's own command line compiled from its sources (oracle/_ref/FamSeq_ref), line by line."""
    import sys
    import tempfile

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import cli_fuzz

    if not os.path.exists(cli_fuzz.REF):
        pytest.skip("oracle/_ref/FamSeq_ref is not built here")
    with tempfile.TemporaryDirectory(prefix="fsfuzz") as tmp:
        for seed in range(first, first + 8):
            for one in (cli_fuzz.run_seed, cli_fuzz.run_seed_lk):  # the vcf driver, then the LK driver on a random table
                line, failed = one(seed, tmp)
                assert not failed, line
