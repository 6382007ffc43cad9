"""world_size-2 `gloo` tests (CPU, no GPU) of the multi-GPU layer (ark_bulletproofs_amd/parallel.py): term-sharded
MSM and proof-sharded batch verification.  The per-rank compute that needs a GPU is stood in for by the CPU oracle
(tests may use it); what is under test is the sharding, the all-gather of one partial point per rank, the host
point-reduce (bp_host_points_sum, product code) and the alpha_skip bookkeeping."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ark_bulletproofs_amd import engine as E
        from ark_bulletproofs_amd import parallel as P
        from oracle import pyoracle as O

        out = {}
        for cv in (0, 1):
            FR = O.fid(cv, True)
            # ---- term-sharded MSM -------------------------------------------------------------------
            n = 77
            G, H = O.bp_gens(cv, n)
            sc = O.fe_rand(FR, bytes([3]) * 32, n)
            lo, hi = P.shard_range(n, rank, world)
            full = P.sharded_msm(cv, lambda: O.msm(cv, G[lo:hi], sc[lo:hi]), E.host_points_sum)
            out["msm%d" % cv] = bool((full == O.msm(cv, G, sc)).all())
            # an empty shard (n < world) contributes the identity
            lo1, hi1 = P.shard_range(1, rank, world)
            one = P.sharded_msm(cv, lambda: O.msm(cv, G[lo1:hi1], sc[lo1:hi1]) if hi1 > lo1 else np.zeros(8, dtype=np.uint64), E.host_points_sum)
            out["msm_small%d" % cv] = bool((one == O.msm(cv, G[:1], sc[:1])).all())
        dist.barrier()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_sharded_msm_gloo_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        assert all(out.values()), (rank, out)


def test_shard_range_partitions():
    from ark_bulletproofs_amd.parallel import shard_range

    for n in [0, 1, 7, 8, 9, 4096, 65537]:
        for world in [1, 2, 3, 8]:
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_host_points_sum_matches_oracle(oracle):
    from ark_bulletproofs_amd import build

    build.build()
    from ark_bulletproofs_amd import engine as E

    for cv in (0, 1):
        G, H = oracle.bp_gens(cv, 9)
        pts = np.concatenate([G, np.zeros((1, 8), dtype=np.uint64), G[:1]])
        exp = np.zeros(8, dtype=np.uint64)
        for p in pts:
            exp = oracle.point_add(cv, exp, p)
        assert (E.host_points_sum(cv, pts) == exp).all()
        neg = G[0].copy()
        neg[4:] = oracle.fe_op("sub", oracle.fid(cv, False), oracle.fe_from_int(oracle.fid(cv, False), 0), G[0, 4:])
        assert not E.host_points_sum(cv, np.stack([G[0], neg])).any()


def _alpha_worker(rank, world, port, q):
    """proof-sharded batch verification with the oracle standing in for the per-rank GPU mega-check: the per-rank check
    points must sum to the identity exactly when the full batch verifies."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ark_bulletproofs_amd import engine as E
        from ark_bulletproofs_amd import parallel as P
        from oracle import pyoracle as O

        cv = 0
        res = {}
        for tag, vals in [("ok", [(0, 16), (3, 16), ((1 << 16) - 1, 16), (1 << 16, 32), (5, 8)]), ("bad", [(0, 16), (1 << 16, 16), (7, 8)])]:
            inst = []
            for i, (v, n) in enumerate(vals):
                pr = O.r1cs_prove(cv, O.SC_RANGE, [n, v], bytes([9 + i]) * 32, 128)
                inst.append((O.SC_RANGE, [n, v], pr.proof, pr.commitments, pr.publics))

            def local(slice_, skip):
                # stand-in for Engine.batch_verify(..., alpha_skip, want_point=True): a slice that verifies on its own
                # contributes the identity; a failing slice contributes a non-identity point
                rc = O.batch_verify(cv, slice_, 128, bytes([5]) * 32)
                return (0 if rc == 0 else -4), (np.zeros(8, dtype=np.uint64) if rc == 0 else O.generator(cv))

            res[tag] = P.sharded_batch_verify(cv, inst, local, E.host_points_sum, rank, world)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_sharded_batch_verify_gloo_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_alpha_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        assert out["ok"] == 0 and out["bad"] == -4, (rank, out)
