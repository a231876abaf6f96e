"""The path at BASELINE.json's own sizes (1 M ped5 sites, 10 M ped10 sites, ped15 at 131,072 sites
per kernel and at its full 1 M sites) on input generated straight into HBM: size-independent properties over the WHOLE
batch — rows are distributions, every status byte 0, bit-reproducible, independent of the site's
position and of how the batch is cut — plus a sample of sites against the CPU oracle.  Everything
goes through the C ABI's device entry point (famseq_bn_batch_device)."""
import numpy as np
import pytest

import famseq_amd as fs
import oracle

pytestmark = pytest.mark.gpu
RTOL = 1e-9  # the north-star bar is 1e-6 relative; observed <= 4e-15

ENGINES = [("team", dict(enum_impl=0)), ("lane", dict(enum_impl=1)), ("elim", dict(engine=fs.ENGINE_ELIM))]


def _run(ctx, torch, n, lk, flags, post, single, status, first=0, count=None):
    """famseq_bn_batch_device over sites [first, first+count) of the resident arrays."""
    count = n - first if count is None else count
    w = lk.shape[1] * 3 * 8
    ctx.bn_batch_device(count, lk.data_ptr() + first * w, flags.data_ptr() + first, post.data_ptr() + first * w,
                        single.data_ptr() + first * w, status.data_ptr() + first, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()


@pytest.mark.parametrize("name,cfg,n_sites,engines,n_oracle", [
    ("ped5", 1, 1_000_000, ("team", "lane", "elim"), 256),     # BASELINE configs[1]
    ("ped10", 2, 10_000_000, ("lane", "elim"), 256),           # configs[2] (and [3] per GPU, weak scaling)
    ("ped10", 2, 1_000_000, ("team",), 256),                   # the compiled-in kernel: 0.1 s per M sites
    ("ped15", 4, 131_072, ("team", "lane", "elim"), 2),        # configs[4] shape: 3^15 configurations per site
    ("ped15", 4, 1_000_000, ("lane", "elim"), 2),              # configs[4] at its full 1 M sites (0.5 s per enumeration launch)
])
def test_properties_at_baseline_size(name, cfg, n_sites, engines, n_oracle):
    import torch

    dev = torch.device("cuda", 0)
    ped = fs.synthetic_pedigree(name)
    n = ped.n
    mo, fa = ped.relations()
    lk, flags = fs.synth.gen_batch_torch(mo.tolist(), fa.tolist(), n_sites, cfg, device=dev)
    idx = np.sort(np.random.RandomState(5).choice(n_sites, n_oracle, replace=False))
    tidx = torch.from_numpy(idx).to(dev)
    ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders).bn_batch(
        lk[tidx].cpu().numpy(), flags[tidx].cpu().numpy(), threads=8)
    perm = torch.randperm(n_sites, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    lk_p, fl_p = lk[perm].contiguous(), flags[perm].contiguous()
    model = fs.make_model(ped)
    by_engine = {}
    for label, opt in ENGINES:
        if label not in engines:
            continue
        ctx = fs.Context(model, **opt)
        post, single = torch.empty_like(lk), torch.empty_like(lk)
        status = torch.full((n_sites,), 77, dtype=torch.uint8, device=dev)
        _run(ctx, torch, n_sites, lk, flags, post, single, status)
        # every site took the full computation and every row is a distribution
        assert int((status != 0).sum()) == 0, label
        assert bool((post >= 0).all()) and float((post.sum(dim=2) - 1).abs().max()) < 1e-12, label
        assert float((single.sum(dim=2) - 1).abs().max()) < 1e-12, label
        # the sample against the oracle: status and single posterior bit-exact, posterior to RTOL
        assert np.array_equal(single[tidx].cpu().numpy(), ref[1]), label
        np.testing.assert_allclose(post[tidx].cpu().numpy(), ref[0], rtol=RTOL, atol=0, err_msg=label)
        # bit-reproducible
        post2, single2 = torch.empty_like(lk), torch.empty_like(lk)
        status2 = torch.empty_like(status)
        _run(ctx, torch, n_sites, lk, flags, post2, single2, status2)
        assert torch.equal(post2, post) and torch.equal(single2, single), label
        # independent of where a site sits in the batch ...
        _run(ctx, torch, n_sites, lk_p, fl_p, post2, single2, status2)
        assert torch.equal(post2, post[perm]) and torch.equal(single2, single[perm]), label
        # ... and of how the batch is cut (ragged cut: not a multiple of any chunk size)
        cut = n_sites // 3 + 1
        post2.zero_()
        _run(ctx, torch, n_sites, lk, flags, post2, single2, status2, 0, cut)
        _run(ctx, torch, n_sites, lk, flags, post2, single2, status2, cut, n_sites - cut)
        assert torch.equal(post2, post), label
        by_engine[label] = post
        ctx.close()
        del post2, single2
    # independent algorithms agree over the whole batch (enumeration vs sum-product)
    if "lane" in by_engine and "elim" in by_engine:
        a, b = by_engine["lane"], by_engine["elim"]
        assert float(((a - b).abs() / a.clamp_min(1e-300)).max()) < 1e-11
