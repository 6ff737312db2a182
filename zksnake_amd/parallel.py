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


_GATHER_BUFFERS = {}   # (world, words, device) -> (pinned host input, device input, device output, pinned host output)


def all_gather_limbs(mine, device=None):
    """every rank contributes a flat uint64 vector of the same length; returns the (world, len) array.
    Uses the default torch.distributed process group (RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    mine = np.ascontiguousarray(mine, dtype=np.uint64).reshape(-1)
    words = mine.shape[0]
    # one flat collective (RCCL all-gather over xGMI; gloo on the CPU).  The output is the concatenation of the ranks'
    # vectors -- the form both backends accept -- and there is no fallback path: a failing collective surfaces as it is
    # (round-2 verdict: a retry with all_gather would have masked a genuine RCCL error on the first multi-GPU run)
    if device is None:
        src = torch.from_numpy(mine.view(np.int64))
        flat = torch.empty(world * words, dtype=torch.int64)
        dist.all_gather_into_tensor(flat, src)
        return flat.view(world, words).numpy().view(np.uint64)
    # On the GPU the payload is a few hundred bytes and the call sits on the critical path of every sharded MSM / proof: the
    # staging tensors (page-locked on the host side) are kept per shape instead of being allocated per call
    key = (world, words, str(device))
    bufs = _GATHER_BUFFERS.get(key)
    if bufs is None:
        bufs = (torch.empty(words, dtype=torch.int64).pin_memory(), torch.empty(words, dtype=torch.int64, device=device),
                torch.empty(world * words, dtype=torch.int64, device=device), torch.empty(world * words, dtype=torch.int64).pin_memory())
        _GATHER_BUFFERS[key] = bufs
    h_in, d_in, d_out, h_out = bufs
    h_in.numpy()[:] = mine.view(np.int64)
    d_in.copy_(h_in, non_blocking=True)
    dist.all_gather_into_tensor(d_out, d_in)
    h_out.copy_(d_out, non_blocking=True)
    torch.cuda.current_stream(device).synchronize()
    return h_out.view(world, words).numpy().view(np.uint64).copy()


def all_gather_sum(curve_id, group, partial, device=None):
    """all ranks contribute one partial point (uint64 limbs) and receive the total"""
    return sum_points(curve_id, group, list(all_gather_limbs(partial, device)))


# ---------------------------------------------------------------------------------------------------------------------
# Task x window partition of a whole Groth16 proof (reference python/zksnake/groth16/protocol.py:133-155:
# prove() is five independent `multiexp` calls).  Window-sharding EVERY MSM on EVERY rank makes each rank pay five sorts, five
# bucket reductions and the whole QAP chain whatever the rank count (round-3 verdict).  Here the five MSMs are laid on one line
# of (task, window) units,
#       <kdelta_1, w>   <tau_1, u>   <tau_1, v>   <tau_2, v>   <target_1, h>
# and the line is cut into `world` contiguous pieces of about equal COST: a rank gets one MSM (or a window range of the G2 one,
# or of <target_1, h>) and evaluates only the part of the QAP its scalars need -- nothing for the witness MSM, one sparse
# product + one inverse transform for u or v, the full chain only for h.  Neighbours on the line share prerequisites (both v
# tasks sit side by side).  The exchange stays ONE all_gather of partial points.

# cost model, milliseconds on one MI355X at n = 2^20 (single-GPU measurements: DESIGN.md section 5, tools/window_range_bench.py
# and the stage timers of the fixed-base plans inside a proof; re-fitted on tools/partition_bench.py, profiles/r04_partition_*).
# `window` (n bucket additions) and the prerequisites scale with n; `fixed` (digits + sort set-up, reduction of the shared
# 2^15-bucket set of 16-bit windows, host tail, launch latencies) does not; `wide` is what the 2^19-bucket set of the 20-bit
# windows adds to the reduction -- a plan over ALL windows of an MSM of 2^20 points or more takes those (13 windows instead of
# 16).  (curve, group) -> (per window, fixed, wide)
_MSM_COST_MS = {
    (0, 1): (0.086, 0.38, 0.20), (0, 2): (0.225, 0.50, 0.35),
    (1, 1): (0.170, 0.50, 0.30), (1, 2): (0.590, 0.90, 0.60),
}
# witness upload, one sparse product + one inverse transform, the whole chain (three products, six transforms)
_QAP_COST_MS = {"upload": 0.60, "u": 0.20, "v": 0.20, "h": 1.05}

PROOF_TASKS = ("k", "u", "v1", "v2", "h")           # line order
TASK_NEEDS = {"k": "w", "u": "u", "v1": "v", "v2": "v", "h": "h"}
TASK_GROUP = {"k": 1, "u": 1, "v1": 1, "v2": 2, "h": 1}


def _segment_cost(curve_id, scale, windows_of, n_windows=None, whole_windows=None, split_is_wide=False):
    """cost of one rank holding `windows_of` = {task: window count > 0}; a task held whole runs in its all-windows layout"""
    if not windows_of:
        return 0.0
    cost = _QAP_COST_MS["upload"] * scale
    needs = {TASK_NEEDS[t] for t in windows_of}
    if "h" in needs:
        cost += _QAP_COST_MS["h"] * scale
    else:
        cost += sum(_QAP_COST_MS[x] * scale for x in ("u", "v") if x in needs)
    for t, k in windows_of.items():
        per_window, fixed, wide = _MSM_COST_MS[(curve_id, TASK_GROUP[t])]
        if split_is_wide:
            fixed += wide
        elif whole_windows and n_windows and k == n_windows[t] and whole_windows.get(t, k) < k:
            k, fixed = whole_windows[t], fixed + wide
        cost += fixed + k * per_window * scale
    return cost


def partition_proof(world, n_windows, curve_id=0, n=1 << 20, whole_windows=None, split_is_wide=False):
    """n_windows: {task: window count of its MSM in the layout of a window-sharded plan} (0 = the MSM does not exist, e.g. no
    private witness); whole_windows: the counts of the all-windows layout (fewer, wider windows from 2^20 points on), used by
    a rank that holds a task whole.  split_is_wide: n_windows already IS that layout (large MSMs: every plan takes the wide
    windows, the bigger bucket set is cheap next to the additions saved).
    Returns (assignment, projected_ms): assignment[rank] = {task: (first, count)} with every window of every task on exactly
    one rank, ranks ordered along the line (count == n_windows[task]: the whole MSM); projected_ms[rank] = the model's cost of
    that rank.  Minimises the slowest rank, then the total (fewer split MSMs)."""
    scale = n / float(1 << 20)
    units = [(t, w) for t in PROOF_TASKS for w in range(int(n_windows.get(t, 0)))]
    U = len(units)

    def seg(i, j):
        """cost and content of units[i:j]"""
        content = {}
        for t, _ in units[i:j]:
            content[t] = content.get(t, 0) + 1
        return _segment_cost(curve_id, scale, content, n_windows, whole_windows, split_is_wide)

    cost = [[0.0] * (U + 1) for _ in range(U + 1)]
    for i in range(U + 1):
        for j in range(i, U + 1):
            cost[i][j] = seg(i, j)
    INF = (float("inf"), float("inf"))
    # best[r][j] = (max, sum) over the first r ranks covering units[:j]
    best = [[INF] * (U + 1) for _ in range(world + 1)]
    cut = [[0] * (U + 1) for _ in range(world + 1)]
    best[0][0] = (0.0, 0.0)
    for r in range(1, world + 1):
        for j in range(U + 1):
            for i in range(j + 1):
                prev = best[r - 1][i]
                if prev[0] == float("inf"):
                    continue
                c = cost[i][j]
                cand = (max(prev[0], c), prev[1] + c)
                if cand < best[r][j]:
                    best[r][j] = cand
                    cut[r][j] = i
    bounds, j = [], U
    for r in range(world, 0, -1):
        i = cut[r][j]
        bounds.append((i, j))
        j = i
    bounds.reverse()
    assignment, projected = [], []
    for i, j in bounds:
        mine = {}
        for t, w in units[i:j]:
            first, count = mine.get(t, (w, 0))
            mine[t] = (first, count + 1)
        assignment.append(mine)
        projected.append(round(cost[i][j], 4))
    return assignment, projected


def window_partition(world, n_windows):
    """the round 1-3 layout, kept for comparison (`Groth16.shard_over_ranks(partition="window")`): every rank takes a window range
    of EVERY MSM and evaluates the whole QAP"""
    out = [dict() for _ in range(world)]
    for t in PROOF_TASKS:
        nw = int(n_windows.get(t, 0))
        if nw:
            for rank, (first, count) in enumerate(window_ranges(nw, world)):
                if count:
                    out[rank][t] = (first, count)
    return out
