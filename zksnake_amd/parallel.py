"""
Multi-GPU sharding of one MSM (north star): the windows of the signed-digit
decomposition are split over the ranks, every rank reduces its windows to ONE partial point
(already weighted by 2^(c*w)), and the partials are exchanged with an all_gather -- EC addition is
not a reduction operator of RCCL, so the "sum" is gather + local point additions.
The payload is one affine point per rank (64..192 bytes).
"""

import numpy as np

from . import _native as N


def window_ranges(n_windows, world):
    """contiguous, balanced split: [(first, count)] per rank (count may be 0 when world > n_windows)"""
    base, extra = divmod(n_windows, world)
    out, first = [], 0
    for rank in range(world):
        count = base + (1 if rank < extra else 0)
        out.append((first, count))
        first += count
    return out


def sum_points(curve_id, group, partials):
    """host sum of affine points given as uint64 limb arrays (one inversion in total)"""
    lib = N.load()
    stack = np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.uint64) for p in partials]))
    out = np.zeros(N.point_limbs(curve_id, group), dtype=np.uint64)
    N.check(lib.zk_point_sum(curve_id, group, stack.shape[0], N.u64p(stack), N.u64p(out)))
    return out


def all_gather_limbs(mine, device=None):
    """every rank contributes a flat uint64 vector of the same length; returns the (world, len) array.
    Uses the default torch.distributed process group (RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    mine = torch.from_numpy(np.ascontiguousarray(mine, dtype=np.uint64).view(np.int64))
    if device is not None:
        mine = mine.to(device)
    # one flat collective (RCCL all-gather over xGMI; gloo on the CPU).  The output is the concatenation of the ranks'
    # vectors -- the form both backends accept -- and there is no fallback path: a failing collective surfaces as it is
    # (round-2 verdict: a retry with all_gather would have masked a genuine RCCL error on the first multi-GPU run)
    flat = torch.zeros(world * mine.numel(), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(flat, mine.reshape(-1))
    gathered = flat.view((world,) + tuple(mine.shape))
    return gathered.cpu().numpy().view(np.uint64)


def all_gather_sum(curve_id, group, partial, device=None):
    """all ranks contribute one partial point (uint64 limbs) and receive the total"""
    return sum_points(curve_id, group, list(all_gather_limbs(partial, device)))
