"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests).  The hot path shards with NO data-path collective; the only exchange is an all-gather of
ONE 64-byte partial point per rank per MSM (group addition is not an RCCL reduction op, so: gather, then a
local (world-1)-add point-reduce).  SURVEY.md §8(e).

  sharded_msm           term sharding: rank r owns terms [lo_r, hi_r) of bases/scalars
  window_sharded_msm    window sharding: every rank holds all terms and owns a range of Pippenger windows
  sharded_batch_verify  whole proofs per rank; by linearity the sum of the per-rank mega-check points is the
                        reference's single MSM (src/r1cs/verifier.rs:685)
  enable_window_sharding  whole prover / verifier calls with every inner MSM window-sharded (bp_ctx_set_window_shard);
                          the exchanges go through host callbacks (any torch.distributed backend: gloo in the CPU tests)
  enable_native_sharding  the same partition with the exchanges as ncclAllGather inside the library (bp_ctx_rccl_init)
  sharded_ipa_create    InnerProductProof::create with every vector partitioned index-cyclically (rank r owns the elements
                        i = r mod world): element i and its fold partner n/2 + i live on the same rank, so all folds are
                        local; a round exchanges one pair of partial (L, R) points per rank
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


def allgather_words(arr, group=None, device=None):
    """all-gather a u64 array of identical shape on every rank -> (world, ...) array"""
    import torch
    import torch.distributed as dist

    a = np.ascontiguousarray(arr, dtype=np.uint64)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return a.reshape((1,) + a.shape)
    t = torch.from_numpy(a.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(outs, t, group=group)
    return np.stack([o.cpu().numpy().view(np.uint64) for o in outs])


def sharded_ipa_create(curve, stepper, Q, G_factors, H_factors, G_vec, H_vec, a_vec, b_vec, challenge, points_sum, rank, world, group=None, device=None,
                       allgather=None):
    """InnerProductProof::create (src/inner_product_proof.rs:37-239) across `world` ranks.  Every rank passes the same full
    vectors (only its cyclic slice is uploaded) and the same `challenge(L, R) -> u` transcript step; every rank returns the same
    (L_vec, R_vec, a, b), bit-identical to the single-GPU result.  `stepper` is an Engine (or anything with ipa_begin /
    ipa_round_LR / ipa_round_fold / ipa_export / ipa_finish).

    Rounds while a rank holds >= 2 elements: local partial L, R (the <a_L, b_R> * Q term is linear too), all-gather of
    (world x 2) points, host point-reduce, challenge, local fold.  When one element per rank is left the world elements are
    gathered (with the scalar factors the engine still owes the generators as G_factors / H_factors of the continuation) and
    every rank finishes the last lg(world) rounds redundantly.  `allgather(arr) -> (world, ...)` replaces the torch.distributed
    exchange (tests drive several ranks as threads of one process)."""
    if allgather is None:
        def allgather(arr):
            return allgather_words(arr, group, device)
    n = len(a_vec)
    if n == 0 or n & (n - 1):
        raise ValueError("sharded_ipa_create: n must be a power of two")   # the reference asserts (:66)
    if world & (world - 1):
        raise ValueError("sharded_ipa_create: world size must be a power of two")
    Ls, Rs = [], []

    def finish_locally(Q, Gf, Hf, G, H, a, b):
        stepper.ipa_begin(Q, Gf, Hf, G, H, a, b)
        m = len(a)
        while m > 1:
            L, R = stepper.ipa_round_LR()
            stepper.ipa_round_fold(challenge(L, R))
            Ls.append(L)
            Rs.append(R)
            m //= 2
        return stepper.ipa_finish()

    if world == 1 or n < 2 * world:
        ao, bo = finish_locally(Q, G_factors, H_factors, G_vec, H_vec, a_vec, b_vec)
        return np.array(Ls).reshape(-1, 8), np.array(Rs).reshape(-1, 8), ao, bo
    sl = slice(rank, None, world)
    stepper.ipa_begin(Q, np.asarray(G_factors)[sl], np.asarray(H_factors)[sl], np.asarray(G_vec)[sl], np.asarray(H_vec)[sl], np.asarray(a_vec)[sl],
                      np.asarray(b_vec)[sl])
    m = n // world
    while m > 1:
        Lp, Rp = stepper.ipa_round_LR()
        parts = allgather(np.stack([Lp, Rp]))          # (world, 2, 8)
        L, R = points_sum(curve, parts[:, 0]), points_sum(curve, parts[:, 1])
        stepper.ipa_round_fold(challenge(L, R))
        Ls.append(L)
        Rs.append(R)
        m //= 2
    a1, b1, G1, H1, gG, gH = stepper.ipa_export(1)
    allv = allgather(np.concatenate([a1[0], b1[0], G1[0], H1[0]]))   # (world, 24): global index = rank
    ao, bo = finish_locally(Q, np.tile(gG, (world, 1)), np.tile(gH, (world, 1)), allv[:, 8:16], allv[:, 16:24], allv[:, 0:4], allv[:, 4:8])
    return np.array(Ls).reshape(-1, 8), np.array(Rs).reshape(-1, 8), ao, bo


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


def enable_window_sharding(engine, curve, points_sum, rank, world, group=None, device=None, allgather=None, cyclic_ipa=True):
    """north_star's partition for one large proof: every rank runs the same prover / verifier call on the same statement, every
    MSM inside accumulates only this rank's Pippenger windows, and the partial points are summed here (all-gather of one
    64-byte point per rank + host point-reduce).  With cyclic_ipa (and a power-of-two world) the inner-product argument is
    partitioned as well: rank r folds only the elements i = r mod world (bp_ctx_set_shard_allgather), the L / R of a round are
    sums of per-rank partial MSMs through the same point-reduce, and the last ~1024 elements are all-gathered once.
    All ranks end with the identical proof."""
    if allgather is None:
        def allgather(arr):
            return allgather_words(arr, group, device)

    def reduce_fn(xy):
        return points_sum(curve, allgather(xy).reshape(-1, 8))

    def gather_bytes(blk):
        return allgather(np.ascontiguousarray(blk).view(np.uint64)).view(np.uint8)     # block sizes are multiples of 32 bytes

    engine.set_window_shard(rank, world, reduce_fn if world > 1 else None)
    engine.set_shard_allgather(gather_bytes if (world > 1 and cyclic_ipa) else None)


def enable_native_sharding(engine, rank, world, group=None, device=None):
    """The same partition with the collectives INSIDE the library (bp_ctx_rccl_init): rank 0 draws an ncclUniqueId, this host
    layer only carries its 128 bytes to the other ranks (one torch.distributed broadcast), and from then on every exchange of
    the sharded prover / verifier — the per-MSM point-reduce and the one vector gather of the index-cyclic inner-product
    argument — is an ncclAllGather on the engine's own HIP stream.  No Python, no interpreter lock on the data path."""
    import torch
    import torch.distributed as dist

    from . import engine as E

    if world <= 1:
        engine.rccl_shutdown()
        return
    buf = np.frombuffer(E.rccl_unique_id(), dtype=np.uint8).copy() if rank == 0 else np.zeros(128, dtype=np.uint8)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=0, group=group)
    engine.rccl_init(t.cpu().numpy().tobytes(), rank, world)


def sharded_batch_verify(curve, instances, local_batch_verify, points_sum, rank, world, group=None, device=None):
    """instances: the FULL ordered instance list (every rank passes the same list); rank r verifies its block with
    alpha_skip = lo_r.  local_batch_verify(slice, alpha_skip) -> (status, check_point).  Returns 0 iff the batch is valid.

    The batch is valid iff EVERY rank's status is 0.  A rank's batch_verify can fail before its mega-check MSM runs (an identity
    A_I1 / T_i / L_j, a truncated L_vec, a malformed proof: src/r1cs/verifier.rs:420-470 return early) and then has no check
    point to contribute, so the point sum alone must never decide: the statuses are max-reduced first.  The sum of the ranks'
    check points (by linearity the reference's single MSM, :685) is kept as a consistency check on top.  The first hard error
    (anything but VerificationError) wins over -4, as the reference's `?` returns the first error."""
    import torch
    import torch.distributed as dist

    lo, hi = shard_range(len(instances), rank, world)
    status, pt = 0, np.zeros(8, dtype=np.uint64)
    if hi > lo:
        status, pt = local_batch_verify(instances[lo:hi], lo)
    if status != 0:
        pt = np.zeros(8, dtype=np.uint64)   # a failed rank's buffer is not a check point
    parts = allgather_points(pt, group, device)
    total = points_sum(curve, parts)
    worst = status
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        # statuses are <= 0; order them so that hard errors beat -4 beats 0: key = 0 (ok), 1 (-4), 2 + |code| (hard error)
        key = 0 if status == 0 else 1 if status == -4 else 2 + abs(int(status))
        flag = torch.tensor([key], dtype=torch.int64)
        if device is not None:
            flag = flag.to(device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        key = int(flag.item())
        worst = 0 if key == 0 else -4 if key == 1 else -(key - 2)
    if worst != 0:
        return worst
    return 0 if not total.any() else -4
