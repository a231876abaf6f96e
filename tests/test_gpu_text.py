"""Device-side text (famseq_bn_call_text_batch, rows N1 + N2): the characters the reference's drivers append to every sample
column — GPP triple, FPP triple, called genotype (file.cpp:696-745) — formatted on the GPU.  Checked against Python's
'%g' (correctly rounded, i.e. what printf and `ostream << double` print) of the numbers famseq_bn_call_batch returns, on
every fixture through three engines, and the formatter alone on ties, decade edges and random Phred-shaped values."""
import math

import numpy as np
import pytest

import famseq_amd as fs
from _cases import load_cases

pytestmark = pytest.mark.gpu
CASES = load_cases()
ENGINES = [("lane, fused", dict(enum_impl=1)), ("sum-product, fused", dict(engine=fs.ENGINE_ELIM)),
           ("team + separate stages", dict(enum_impl=0))]


def expected_record(g, f, gt):
    return ("%g,%g,%g:%g,%g,%g:%s\t" % (g[0], g[1], g[2], f[0], f[1], f[2], {0: "0/0", 1: "0/1"}.get(int(gt), "1/1"))).encode()


@pytest.fixture(scope="module")
def any_ctx():
    c = next(x for x in CASES if x.name == "bn_synth:quad")
    ctx = fs.Context(fs.make_model(c.pedigree(), **c.consts))
    yield ctx
    ctx.close()


def test_device_formatter_equals_printf(any_ctx):
    rng = np.random.RandomState(7)
    vals = [0.0, 1.0, 10.0, 99999.0, 100000.0, 999999.0, 999999.4, 524288.0, 745679.5, 1e-4, 9.99999e-5, 9.999995e-5, 0.000123456,
            1e-5, 1e-16, 1.00000001e-16, 4.8216e-16, 0.5, 1.5, 2.5, 123456.5, 123457.5, 12345.65, 2.83856e-06, 61.8469, 220.877,
            3239.9999, 0.1, 0.2, 0.3, 1.0 / 3, 2.0 / 3]
    # exact ties at every digit position, one ulp either side of powers of ten and of six-digit decimals
    for _ in range(20000):
        t = float(rng.randint(100000, 1000000)) + 0.5
        vals.append(t)
        vals.extend(math.ldexp(t, -j) for j in range(1, 41, 3))
    for x in range(-16, 6):
        p = 10.0 ** x
        vals.extend([p, np.nextafter(p, 0), np.nextafter(p, 1e9)])
        for _ in range(300):
            d = float(rng.randint(100000, 1000000)) * 10.0 ** (x - 5)
            h = (float(rng.randint(100000, 1000000)) + 0.5) * 10.0 ** (x - 5)
            vals.extend([d, np.nextafter(d, 0), np.nextafter(d, 1e9), h, np.nextafter(h, 0), np.nextafter(h, 1e9)])
    vals = np.array([v for v in vals if v == 0 or 1e-16 <= v < 999999.5])
    # log-uniform over the whole range, and Phred values of random probabilities (next to 1, uniform, tiny)
    u = rng.random_sample(400000)
    vals = np.concatenate([vals, 10.0 ** (-16 + 21.9 * u), np.abs(-10 * np.log10(rng.random_sample(400000))),
                           np.abs(-10 * np.log10(1 - np.ldexp(rng.random_sample(200000), -rng.randint(0, 53, 200000)))),
                           np.abs(-10 * np.log10(10.0 ** (-300 * rng.random_sample(200000))))])
    vals = vals[(vals == 0) | ((vals >= 1e-16) & (vals < 999999.5))]
    got = any_ctx.g6_probe(vals)
    bad = [(v, g) for v, g in zip(vals, got) if g != ("%g" % v).encode()]
    assert not bad, bad[:10]
    assert any_ctx.g6_probe([float("nan"), -1.0, 1e7, 1e-20]) == [b"nan"] * 4  # outside the domain: never produced for a computed site


@pytest.mark.parametrize("label,opts", ENGINES, ids=[e[0] for e in ENGINES])
def test_text_records_on_every_fixture(label, opts):
    n = 0
    for c in CASES:
        if not c.name.startswith("bn_"):
            continue
        ctx = fs.Context(fs.make_model(c.pedigree(), **c.consts), **opts)
        seq = np.nonzero(c.sequenced)[0][::-1].copy()  # a column order different from PED order
        gpp, fpp, fgt, st = ctx.bn_call_batch(seq, lk=c.lk, flags=c.flags)
        text, st2 = ctx.bn_call_text_batch(seq, lk=c.lk, flags=c.flags)
        ctx.close()
        assert np.array_equal(st, st2), c.name
        for s in np.nonzero((st & 3) == 0)[0]:
            for j in range(len(seq)):
                rec = text[s, j]
                assert bytes(rec[:rec[-1]]) == expected_record(gpp[s, j], fpp[s, j], fgt[s, j]), (c.name, s, j)
                assert not rec[rec[-1]:-1].any()
                n += 1
    assert n > 1000


def test_text_from_packed_pls_at_size():
    """A batch of several chunks and tiles (the text kernel walks 256 pairs per tile; the host pipeline cuts chunks)."""
    ped = fs.synthetic_pedigree("ped10")
    s = 70001
    mo, fa = ped.relations()
    pl, known, _ = fs.synth.gen_sites(mo, fa, s, seed=fs.synth.SEED_BASE + 2)
    seq = np.arange(ped.n, dtype=np.int32)
    ctx = fs.Context(fs.make_model(ped), engine=fs.ENGINE_ELIM, chunk_sites=16384)
    gpp, fpp, fgt, st = ctx.bn_call_batch(seq, pl16=pl.astype(np.uint16), flags=known.astype(np.uint8))
    text, st2 = ctx.bn_call_text_batch(seq, pl16=pl.astype(np.uint16), flags=known.astype(np.uint8))
    ctx.close()
    assert np.array_equal(st, st2) and not (st & 3).any()
    for i in list(range(0, s, 977)) + [16383, 16384, s - 1]:
        for j in range(ped.n):
            rec = text[i, j]
            assert bytes(rec[:rec[-1]]) == expected_record(gpp[i, j], fpp[i, j], fgt[i, j]), (i, j)
    # every record well-formed: a length, a tab at its end, zeros behind it
    ln = text[:, :, -1].astype(int)
    assert ln.min() >= 16 and ln.max() <= 76
    flat = text.reshape(-1, fs.TEXT_STRIDE)
    assert np.all(flat[np.arange(len(flat)), ln.ravel() - 1] == ord("\t"))


@pytest.mark.parametrize("label,opts", ENGINES, ids=[e[0] for e in ENGINES])
def test_device_resident_call_path_gives_the_host_entry_bits(label, opts):
    """famseq_bn_call_batch_device on resident buffers (packed PLs and fp64 rows; with every output, and with the text alone)
    against famseq_bn_call_batch / famseq_bn_call_text_batch on host buffers: the same kernels, the same bits."""
    import torch

    dev = torch.device("cuda", 0)
    ped = fs.synthetic_pedigree("ped10")
    s = 5000 if opts.get("enum_impl") == 0 else 40000
    mo, fa = ped.relations()
    pl, known, _ = fs.synth.gen_sites(mo, fa, s, seed=fs.synth.SEED_BASE + 2)
    pl16 = pl.astype(np.uint16)
    pl16[5, 3] = 0xFFFF   # a missing sample
    pl16[7, 2, 1] = 3300  # beyond where pow(10, -k / 10) is 0 in double
    pl16[9, 1, 2] = 1500  # beyond the part of the table the sum-product form keeps in LDS
    flags = known.astype(np.uint8)
    seq = np.arange(ped.n, dtype=np.int32)[::-1].copy()
    ctx = fs.Context(fs.make_model(ped), **opts)
    gpp, fpp, fgt, st = ctx.bn_call_batch(seq, pl16=pl16, flags=flags)
    text, _ = ctx.bn_call_text_batch(seq, pl16=pl16, flags=flags)
    d_pl, d_fl = torch.from_numpy(pl16.view(np.int16)).to(dev), torch.from_numpy(flags).to(dev)
    d_gpp, d_fpp = torch.empty((s, ped.n, 3), dtype=torch.float64, device=dev), torch.empty((s, ped.n, 3), dtype=torch.float64, device=dev)
    d_fgt, d_st = torch.empty((s, ped.n), dtype=torch.int8, device=dev), torch.empty(s, dtype=torch.uint8, device=dev)
    d_text = torch.zeros((s, ped.n, fs.TEXT_STRIDE), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.bn_call_batch_device(s, seq, d_pl16=d_pl.data_ptr(), d_flags=d_fl.data_ptr(), d_gpp=d_gpp.data_ptr(), d_fpp=d_fpp.data_ptr(),
                             d_fgt=d_fgt.data_ptr(), d_status=d_st.data_ptr(), d_text=d_text.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_st.cpu().numpy(), st) and np.array_equal(d_fgt.cpu().numpy(), fgt)
    assert np.array_equal(d_gpp.cpu().numpy(), gpp, equal_nan=True) and np.array_equal(d_fpp.cpu().numpy(), fpp, equal_nan=True)
    assert np.array_equal(d_text.cpu().numpy(), text)
    # the text alone (the numbers go through this context's scratch), from fp64 rows this time
    lk = fs.synth.pl_to_lk(pl)[:2000]
    t2, st2 = ctx.bn_call_text_batch(seq, lk=lk, flags=flags[:2000])
    d_lk = torch.from_numpy(np.ascontiguousarray(lk)).to(dev)
    d_text.zero_()
    ctx.bn_call_batch_device(2000, seq, d_lk=d_lk.data_ptr(), d_flags=d_fl.data_ptr(), d_text=d_text.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_text[:2000].cpu().numpy(), t2)
    ctx.close()


def test_phase_clock_measuring_aid(tmp_path):
    """FAMSEQ_PHASE_CLOCK=1 (how the call-path kernel's phases were weighed: DESIGN.md 2.4) generates another source — marks and
    an extra argument — in its own cache; it must keep compiling, report shares that sum to 100 %, and change no result."""
    import os
    import subprocess
    import sys

    code = r'''
import numpy as np, famseq_amd as fs
ped = fs.synthetic_pedigree("ped10")
mo, fa = ped.relations()
pl, known, _ = fs.synth.gen_sites(mo, fa, 3000, seed=fs.synth.SEED_BASE + 2)
ctx = fs.Context(fs.make_model(ped), engine=fs.ENGINE_ELIM)
out = ctx.bn_call_batch(np.arange(ped.n, dtype=np.int32), pl16=pl.astype(np.uint16), flags=known.astype(np.uint8))
np.savez(__import__("sys").argv[1], *out)
'''
    outs = []
    for clock in ("0", "1"):
        env = dict(os.environ, FAMSEQ_KERNEL_CACHE=str(tmp_path / ("kc" + clock)))
        os.makedirs(env["FAMSEQ_KERNEL_CACHE"])
        if clock == "1":
            env["FAMSEQ_PHASE_CLOCK"] = "1"
        f = str(tmp_path / ("o%s.npz" % clock))
        p = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, env=env, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert p.returncode == 0, p.stderr[-2000:]
        if clock == "1":
            line = [x for x in p.stderr.splitlines() if "famseq phase clock" in x][-1]
            shares = [float(x.split("]")[1].split("%")[0]) for x in line.split("[")[1:]]
            assert abs(sum(shares) - 100) < 1.0 and shares[0] > 0 and shares[3] > 0, line
        else:
            assert "phase clock" not in p.stderr
        z = np.load(f)
        outs.append([z[k] for k in z.files])
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)
