"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests).  The hot path shards with NO data-path collective; the only exchange is an all-gather of
ONE 64-byte partial point per rank per MSM (group addition is not an RCCL reduction op, so: gather, then a
local (world-1)-add point-reduce).  SURVEY.md §8(e).

  sharded_msm           term sharding: rank r owns terms [lo_r, hi_r) of bases/scalars
  window_sharded_msm    window sharding: every rank holds all terms and owns a range of Pippenger windows
  sharded_batch_verify  whole proofs per rank; by linearity the sum of the per-rank mega-check points is the
                        reference's single MSM (src/r1cs/verifier.rs:685)
"""
import numpy as np


def shard_range(n, rank, world):
    """contiguous block partition of n units; sizes differ by at most one"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_points(point_xy, group=None, device=None):
    """all-gather one affine point (8 x u64) per rank -> (world, 8) array, identical on every rank"""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.asarray(point_xy, dtype=np.uint64).reshape(1, 8)
    t = torch.from_numpy(np.ascontiguousarray(point_xy, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(outs, t, group=group)
    return np.stack([o.cpu().numpy().view(np.uint64) for o in outs])


def sharded_msm(curve, local_msm, points_sum, group=None, device=None):
    """local_msm() -> this rank's partial MSM (affine, 8 x u64); returns the full MSM value on every rank"""
    parts = allgather_points(local_msm(), group, device)
    return points_sum(curve, parts)


def window_sharded_msm(curve, n, local_msm_windows, window_count, points_sum, rank, world, group=None, device=None):
    """Window sharding (every rank holds all bases and scalars): rank r accumulates Pippenger windows [lo_r, hi_r) of the
    window_count(curve, n)[0] windows; local_msm_windows(lo, hi) -> that partial, already weighted by 2^(c*w)."""
    W, _ = window_count(curve, n)
    lo, hi = shard_range(W, rank, world)
    part = local_msm_windows(lo, hi) if hi > lo else np.zeros(8, dtype=np.uint64)
    return points_sum(curve, allgather_points(part, group, device))


def sharded_batch_verify(curve, instances, local_batch_verify, points_sum, rank, world, group=None, device=None):
    """instances: the FULL ordered instance list (every rank passes the same list); rank r verifies its block with
    alpha_skip = lo_r.  local_batch_verify(slice, alpha_skip) -> (status, check_point).  Returns 0 iff the batch is valid.
    A rank with an error other than VerificationError (malformed proof, missing generators) fails the batch."""
    import torch
    import torch.distributed as dist

    lo, hi = shard_range(len(instances), rank, world)
    status, pt = 0, np.zeros(8, dtype=np.uint64)
    if hi > lo:
        status, pt = local_batch_verify(instances[lo:hi], lo)
    hard_error = status not in (0, -4)
    parts = allgather_points(pt, group, device)
    total = points_sum(curve, parts)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        flag = torch.tensor([1 if hard_error else 0], dtype=torch.int64)
        if device is not None:
            flag = flag.to(device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        hard_error = bool(flag.item())
    if hard_error:
        return status if status not in (0, -4) else -4
    return 0 if not total.any() else -4
