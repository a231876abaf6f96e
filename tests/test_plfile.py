"""Packed PL file format: `FamSeq pack` (host only, no GPU needed) writes what the vcf driver
would compute, Python reads/writes the same bytes."""
import os
import subprocess

import numpy as np
import pytest

from famseq_amd import plfile
from famseq_amd.pedigree import read_ped

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bin", "FamSeq")
TD = os.path.join(ROOT, "tests", "golden", "testdata")


def python_sites(vcf, ped):
    """Integer PLs of the computable sites, by the reference's site rules (file.cpp:362-555)."""
    names, out_flags, out_pl = [], [], []
    for line in open(vcf):
        line = line.rstrip("\n")
        if len(line) < 2:
            break
        if line.startswith("##"):
            continue
        t = line.split("\t")
        if line.startswith("#CHROM"):
            cols = [i for i, c in enumerate(t[9:]) if c in ped.names]
            names = [t[9 + i] for i in cols]
            continue
        if t[3] in (".", "-") or len(t[3]) != 1 or len(t[4]) != 1 or t[0] in ("Y", "chrY", "MT"):
            continue
        c = t[0][3:] if t[0].startswith("chr") else t[0]
        is_x = t[0] in ("X", "chrX", "CHRX")
        if not ((c.isdigit() and 0 < int(c) < 23) or is_x):
            continue
        fmt = t[8].split(":")
        if all(len(t[9 + i]) < 5 for i in cols):
            continue
        ipl = max([k for k, key in enumerate(fmt) if key in ("PL", "GL")], default=-1)
        if ipl < 0:
            continue
        row, integral = [], True
        for i in cols:
            f = t[9 + i]
            sub = f.split(":")
            if len(f) < 5 or len(sub) != len(fmt):
                row.append([0xFFFF] * 3)
                continue
            vals = sub[ipl].split(",")[:3]
            if not all(v.isdigit() for v in vals):
                integral = False
                break
            row.append([min(int(v), 65534) for v in vals])
        if integral:
            out_flags.append((1 if t[2] != "." else 0) | (2 if is_x else 0))
            out_pl.append(row)
    return names, np.array(out_flags, np.uint8), np.array(out_pl, np.uint16)


def test_pack_matches_python_site_rules(tmp_path):
    for vcf, ped in (("test_subset.vcf", "fam01.ped"), ("test_subset.vcf", "fam04.ped"), ("probe.vcf", "probe.ped")):
        out = tmp_path / "x.fspl"
        p = subprocess.run([CLI, "pack", "-vcfFile", os.path.join(TD, vcf), "-pedFile", os.path.join(TD, ped),
                            "-output", str(out)], capture_output=True, text=True)
        assert p.returncode == 0, p.stdout + p.stderr
        names, flags, pl = plfile.read_plfile(str(out))
        want = python_sites(os.path.join(TD, vcf), read_ped(os.path.join(TD, ped)))
        assert names == want[0]
        assert np.array_equal(flags, want[1]) and np.array_equal(pl, want[2])
        if vcf == "probe.vcf":
            assert "skipped" in p.stdout  # the GL line is not packable
            assert int(np.sum(np.all(pl == 0xFFFF, axis=2))) >= 1  # the missing sample at POS 600
        if ped == "fam01.ped":
            assert pl.max() == 65534  # test.vcf holds a PL of 84692 for ind03: clamped (exactly 0 either way)


def _pack(vcf, out, env=None, stdin=None):
    p = subprocess.run([CLI, "pack", "-vcfFile", str(vcf), "-pedFile", TD + "/fam01.ped", "-output", str(out)], capture_output=True,
                       text=True, env=env, stdin=stdin)
    assert p.returncode == 0, p.stdout + p.stderr
    return open(out, "rb").read()


def test_input_shapes_the_line_reader_must_take(tmp_path):
    """The vcf driver cuts lines out of the memory-mapped file (std::getline's lines): the last line counts without a
    newline, processing stops at the first empty line as in the reference (file.cpp: `line.size() < 2`), what cannot be
    mapped (a pipe) is read whole, and the bytes do not depend on how many threads parse a block or how long a block is.
    TestData/test.vcf's body six times over, so that a block has more than the 2,048 lines one thread takes alone."""
    import gzip

    lines = gzip.open(TD + "/test_full.vcf.gz", "rt").read().split("\n")
    head = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if l and not l.startswith("#")]
    text = "\n".join(head + body * 6) + "\n"
    full = tmp_path / "full.vcf"
    full.write_text(text)
    want = _pack(full, tmp_path / "full.fspl")
    hdr, flags, pl = plfile.read_plfile(tmp_path / "full.fspl")
    assert len(flags) == 6 * 12  # the 12 computable sites of test.vcf for fam01, each time
    # no newline after the last line
    (tmp_path / "nonl.vcf").write_text(text[:-1])
    assert _pack(tmp_path / "nonl.vcf", tmp_path / "nonl.fspl") == want
    # one thread, short blocks
    env = dict(os.environ, FAMSEQ_THREADS="1", FAMSEQ_BATCH="97")
    assert _pack(full, tmp_path / "serial.fspl", env=env) == want
    # through a pipe: nothing to map
    with open(full, "rb") as f:
        p = subprocess.Popen(["cat"], stdin=f, stdout=subprocess.PIPE)
        got = _pack("/dev/stdin", tmp_path / "pipe.fspl", stdin=p.stdout)
        p.wait()
    assert got == want
    # an empty line ends the input: only what stands before it is packed
    n_before = len(head) + len(body) * 2
    cut = text.split("\n")
    (tmp_path / "cut.vcf").write_text("\n".join(cut[:n_before] + [""] + cut[n_before:]))
    _pack(tmp_path / "cut.vcf", tmp_path / "cut.fspl")
    assert len(plfile.read_plfile(tmp_path / "cut.fspl")[1]) == 2 * 12


def test_python_writer_roundtrip(tmp_path):
    rng = np.random.RandomState(2)
    pl = rng.randint(0, 65536, size=(1000, 5, 3)).astype(np.uint16)
    flags = rng.randint(0, 4, 1000).astype(np.uint8)
    path = str(tmp_path / "r.fspl")
    plfile.write_plfile(path, ["s%d" % i for i in range(5)], flags, pl)
    for mm in (True, False):
        n, f, p = plfile.read_plfile(path, mmap=mm)
        assert n == ["s0", "s1", "s2", "s3", "s4"] and np.array_equal(f, flags) and np.array_equal(p, pl)
    assert os.path.getsize(path) == 24 + 5 * 32 + 1000 * 31


def test_packed_result_file_round_trip(tmp_path):
    from famseq_amd import plfile

    rng = np.random.RandomState(3)
    s, k = 1000, 4
    st = rng.randint(0, 3, s).astype(np.uint8)
    gpp, fpp = rng.rand(s, k, 3) * 200, rng.rand(s, k, 3) * 200
    fgt = rng.randint(-1, 3, (s, k)).astype(np.int8)
    path = str(tmp_path / "r.fspo")
    plfile.write_results(path, ["a", "b", "c", "d"], st, gpp, fpp, fgt, block=300)  # 4 blocks, the last short
    r = plfile.read_results(path)
    assert r["names"] == ["a", "b", "c", "d"]
    assert np.array_equal(r["status"], st) and np.array_equal(r["fgt"], fgt)
    assert np.array_equal(r["gpp"], gpp) and np.array_equal(r["fpp"], fpp)
    with open(path, "r+b") as f:  # a header that promises more sites than the blocks hold
        f.seek(16)
        f.write(np.array([s + 1], "<u8").tobytes())
    with pytest.raises(ValueError):
        plfile.read_results(path)
