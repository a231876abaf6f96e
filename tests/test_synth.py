"""Seeded synthetic generator: determinism, shard-ability, no shortcut
sites, numpy and torch implementations bit-identical."""
import numpy as np
import pytest

from famseq_amd import synth
from famseq_amd.pedigree import synthetic_pedigree


def test_splitmix64_known_answer():
    # first outputs of SplitMix64(seed=1234567) (public reference values)
    z = synth.splitmix64_at(1234567, np.arange(3, dtype=np.uint64))
    assert z.tolist() == [6457827717110365317, 3203168211198807973, 9817491932198370423]


@pytest.mark.parametrize("name,cfg", [("ped5", 1), ("ped10", 2), ("ped15", 5)])
def test_generator_is_counter_based(name, cfg):
    mo, fa = synthetic_pedigree(name).relations()
    lk, fl = synth.gen_batch(mo, fa, 1000, cfg)
    lk2, fl2 = synth.gen_batch(mo, fa, 300, cfg, first_site=500)
    assert np.array_equal(lk[500:800], lk2) and np.array_equal(fl[500:800], fl2)
    pl, known, geno = synth.gen_sites(mo, fa, 1000, synth.SEED_BASE + cfg)
    assert pl.min() == 0 and pl.max() <= 255
    assert np.all((pl == 0).sum(axis=2) == 1)  # exactly one PL is 0: the true genotype
    # no site can take the -LRC shortcut: that needs both non-zero PLs >= 160 for every member
    second = np.sort(pl, axis=2)[:, :, 1]
    assert second.max() <= 90 and second.min() >= 3
    assert 0.05 < known.mean() < 0.15
    # children are Mendelian-consistent with their parents (gene dropping, no mutation)
    for i in range(len(mo)):
        if mo[i] >= 0:
            bad = ((geno[:, mo[i]] == 0) & (geno[:, fa[i]] == 0) & (geno[:, i] != 0)) | \
                  ((geno[:, mo[i]] == 2) & (geno[:, fa[i]] == 2) & (geno[:, i] != 2))
            assert not bad.any()


def test_torch_generator_matches_numpy():
    torch = pytest.importorskip("torch")
    mo, fa = synthetic_pedigree("ped10").relations()
    lk, fl = synth.gen_batch(mo, fa, 5000, 2, first_site=123)
    tlk, tfl = synth.gen_batch_torch(mo.tolist(), fa.tolist(), 5000, 2, first_site=123, chunk=1700)
    assert np.array_equal(tlk.numpy().view(np.uint64), lk.view(np.uint64))
    assert np.array_equal(tfl.numpy(), fl)


def test_vcf_text_roundtrip(tmp_path):
    ped = synthetic_pedigree("ped5")
    mo, fa = ped.relations()
    pl, known, geno = synth.gen_sites(mo, fa, 20, synth.SEED_BASE + 1)
    path = tmp_path / "s.vcf"
    synth.write_vcf(str(path), ped.names, pl, known, geno)
    rows = [l.rstrip("\n").split("\t") for l in open(path) if not l.startswith("#")]
    assert len(rows) == 20 and rows[3][1] == "4"
    got = np.array([[[int(x) for x in f.split(":")[1].split(",")] for f in r[9:]] for r in rows])
    assert np.array_equal(got, pl)
    assert [r[2] != "." for r in rows] == known.tolist()
