"""Site sharding across GPUs (one process per GPU, torch.distributed; RCCL on ROCm).

Sites are independent (the reference zeroes its state per site, family.cpp:791), so the path
shards with NO data-path collective: rank r owns the contiguous range site_range(S, r, W)
and writes its own slice of the output.  The only optional exchange is an output gather for
a consumer that wants every posterior resident on every rank (gather_sites)."""
import torch
import torch.distributed as dist


def site_range(n_sites, rank, world):
    """Contiguous [lo, hi) of rank `rank`; ranges tile [0, n_sites) in rank order."""
    return n_sites * rank // world, n_sites * (rank + 1) // world


def gather_sites(local, n_sites, group=None):
    """all_gather of per-rank site slices (possibly ragged by one) -> [n_sites, ...] on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = site_range(n_sites, rank, world)
    assert local.shape[0] == hi - lo, "local slice does not match this rank's site range"
    width = max(site_range(n_sites, r, world)[1] - site_range(n_sites, r, world)[0] for r in range(world))
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: hi - lo] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = []
    for r in range(world):
        a, b = site_range(n_sites, r, world)
        out.append(parts[r][: b - a])
    return torch.cat(out, dim=0)


def max_over_ranks(value, device="cpu", group=None):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
