"""N > 1 path on CPU: two and eight gloo ranks shard the seeded site stream, each computes its own
range (the oracle stands in for the GPU here — this test is about the sharding/gather
logic), and the gathered result equals the single-process result."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from famseq_amd.shard import site_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    import oracle
    from famseq_amd import synth, pedigree
    from famseq_amd.shard import site_range, gather_sites, max_over_ranks
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    S = 301
    ped = pedigree.synthetic_pedigree("ped5")
    mo, fa = ped.relations()
    lo, hi = site_range(S, rank, world)
    lk, fl = synth.gen_batch(mo, fa, hi - lo, 1, first_site=lo)   # this rank's own range of the stream
    post, single, st = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders).bn_batch(lk, fl)
    full = gather_sites(torch.from_numpy(post), S)
    stat = gather_sites(torch.from_numpy(st), S)
    t = max_over_ranks(1.0 + rank)
    if rank == 0:
        np.save(sys.argv[1], full.numpy())
        assert t == float(world), t
        assert stat.shape == (S,)
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_site_ranges_tile():
    for s in (0, 1, 7, 10_000_000):
        for w in (1, 2, 3, 8):
            r = [site_range(s, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == s
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_ranks_shard_and_gather(world, tmp_path):
    """world = 8 is the driver's SCALE shape (one rank per GPU of the node): ragged ranges (301 sites over 8 ranks: 37 or 38
    each), the gather's padding and order, the max-over-ranks reduction bench.py's timing uses."""
    import oracle
    from famseq_amd import pedigree, synth

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "full.npy"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), str(out)],
                          env=env, timeout=300)
    ped = pedigree.synthetic_pedigree("ped5")
    mo, fa = ped.relations()
    lk, fl = synth.gen_batch(mo, fa, 301, 1)
    ref = oracle.OracleModel(ped.ids, ped.mids, ped.fids, ped.genders).bn_batch(lk, fl)[0]
    assert np.array_equal(np.load(out), ref)
