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

            def local_early(slice_, skip):
                # what the engine returns when a shard fails BEFORE its mega-check MSM (identity A_I1, truncated L_vec, ...):
                # VerificationError with an all-zero check-point buffer.  The sum of the points is then the identity although
                # the batch is invalid: the statuses must decide.
                rc = O.batch_verify(cv, slice_, 128, bytes([5]) * 32)
                return (0 if rc == 0 else -4), np.zeros(8, dtype=np.uint64)

            res[tag + "_early"] = P.sharded_batch_verify(cv, inst, local_early, E.host_points_sum, rank, world)

            def local_hard(slice_, skip):
                # a malformed proof on the last rank only (FormatError = -6): the hard error wins on every rank
                return (-6 if rank == world - 1 else 0), np.zeros(8, dtype=np.uint64)

            res[tag + "_hard"] = P.sharded_batch_verify(cv, inst, local_hard, E.host_points_sum, rank, world)
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
        assert out["ok_early"] == 0 and out["bad_early"] == -4, (rank, out)   # a failed rank with an identity point fails the batch
        assert out["ok_hard"] == -6 and out["bad_hard"] == -6, (rank, out)


class _OracleIpaStepper:
    """Stand-in for Engine's stepping interface (ipa_begin .. ipa_finish) built from the oracle's field / group primitives,
    following src/inner_product_proof.rs:70-237 round by round; lets the index-cyclic partition run without a GPU."""

    def __init__(self, O, cv):
        self.O, self.cv, self.F = O, cv, O.fid(cv, True)

    def _mul(self, x, y):
        return self.O.fe_op("mul", self.F, x, y)

    def ipa_begin(self, Q, Gf, Hf, G, H, a, b):
        self.Q = np.array(Q, dtype=np.uint64)
        self.Gf, self.Hf = [np.array(x, dtype=np.uint64) for x in Gf], [np.array(x, dtype=np.uint64) for x in Hf]
        self.G, self.H = [np.array(x, dtype=np.uint64) for x in G], [np.array(x, dtype=np.uint64) for x in H]
        self.a, self.b = [np.array(x, dtype=np.uint64) for x in a], [np.array(x, dtype=np.uint64) for x in b]
        self.first = True

    def _ip(self, x, y):
        acc = self.O.fe_from_int(self.F, 0)
        for p, q in zip(x, y):
            acc = self.O.fe_op("add", self.F, acc, self._mul(p, q))
        return acc

    def ipa_round_LR(self):
        O, cv, n = self.O, self.cv, len(self.a) // 2
        one = O.fe_from_int(self.F, 1)
        gf = self.Gf if self.first else [one] * (2 * n)
        hf = self.Hf if self.first else [one] * (2 * n)
        aL, aR, bL, bR = self.a[:n], self.a[n:], self.b[:n], self.b[n:]
        cL, cR = self._ip(aL, bR), self._ip(aR, bL)
        sL = [self._mul(aL[i], gf[n + i]) for i in range(n)] + [self._mul(bR[i], hf[i]) for i in range(n)] + [cL]
        sR = [self._mul(aR[i], gf[i]) for i in range(n)] + [self._mul(bL[i], hf[n + i]) for i in range(n)] + [cR]
        L = O.msm(cv, np.array(self.G[n:] + self.H[:n] + [self.Q]), np.array(sL))
        R = O.msm(cv, np.array(self.G[:n] + self.H[n:] + [self.Q]), np.array(sR))
        return L, R

    def ipa_round_fold(self, u):
        O, cv, n = self.O, self.cv, len(self.a) // 2
        ui = O.fe_op("inv", self.F, u)
        one = O.fe_from_int(self.F, 1)
        gf = self.Gf if self.first else [one] * (2 * n)
        hf = self.Hf if self.first else [one] * (2 * n)
        add = lambda x, y: O.fe_op("add", self.F, x, y)   # noqa: E731
        a2 = [add(self._mul(self.a[i], u), self._mul(ui, self.a[n + i])) for i in range(n)]
        b2 = [add(self._mul(self.b[i], ui), self._mul(u, self.b[n + i])) for i in range(n)]
        G2 = [O.point_add(cv, O.scalar_mul(cv, self.G[i], self._mul(ui, gf[i])), O.scalar_mul(cv, self.G[n + i], self._mul(u, gf[n + i]))) for i in range(n)]
        H2 = [O.point_add(cv, O.scalar_mul(cv, self.H[i], self._mul(u, hf[i])), O.scalar_mul(cv, self.H[n + i], self._mul(ui, hf[n + i]))) for i in range(n)]
        self.a, self.b, self.G, self.H, self.first = a2, b2, G2, H2, False

    def ipa_export(self, n_max):
        one = self.O.fe_from_int(self.F, 1)
        if self.first:
            raise RuntimeError("export before the first fold is not needed by the partition")
        return np.array(self.a), np.array(self.b), np.array(self.G), np.array(self.H), one, one

    def ipa_finish(self):
        assert len(self.a) == 1
        return self.a[0], self.b[0]


def _ipa_worker(rank, world, port, q):
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
        from test_oracle_protocol import _ipa_instance

        out = {}
        for cv in (0, 1):
            for n in (8, 2, 1):   # 8: two partitioned rounds + a gathered one; 2 and 1: fewer elements than 2 * world -> local
                G, H, Q, a, b, Gf, Hf, _ = _ipa_instance(O, cv, n)
                tr = O.Transcript(b"innerproducttest")
                Lo, Ro, ao, bo = O.ipa_create(cv, tr.clone(), Q, Gf, Hf, G, H, a, b)
                t1 = tr.clone()
                t1.append_message(b"dom-sep", b"ipp v1")
                t1.append_u64(b"n", n)

                def ch(L, R, t1=t1, cv=cv):
                    t1.append_point(cv, b"L", L)
                    t1.append_point(cv, b"R", R)
                    return t1.challenge_scalar(cv, b"u")

                L, R, ag, bg = P.sharded_ipa_create(cv, _OracleIpaStepper(O, cv), Q, Gf, Hf, G, H, a, b, ch, E.host_points_sum, rank, world)
                lg = max(n.bit_length() - 1, 0)
                ok = (ag == ao).all() and (bg == bo).all() and L.shape == (lg, 8)
                if lg:
                    ok = ok and (L == Lo).all() and (R == Ro).all()
                out["ipa%d_%d" % (cv, n)] = bool(ok)
        dist.barrier()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_sharded_ipa_create_gloo_world2():
    """index-cyclic IPA over 2 gloo ranks: partial L/R all-gather + host point-reduce, gathered tail rounds"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ipa_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        assert all(out.values()), (rank, out)


def _wshard_worker(rank, world, port, q):
    """enable_window_sharding over gloo: the reduce function installed on the engine must turn each rank's partial point into
    the same full sum on every rank (the engine side — windows per rank inside every MSM — is covered by the GPU tests)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ark_bulletproofs_amd import engine as E
        from ark_bulletproofs_amd import parallel as P
        from oracle import pyoracle as O

        class FakeEngine:
            def set_window_shard(self, r, w, fn=None):
                self.args = (r, w, fn)

            def set_shard_allgather(self, fn=None):
                self.gather = fn

        out = {}
        for cv in (0, 1):
            G, _ = O.bp_gens(cv, world + 1)
            fe = FakeEngine()
            P.enable_window_sharding(fe, cv, E.host_points_sum, rank, world)
            r, w, fn = fe.args
            full = np.zeros(8, dtype=np.uint64)
            for i in range(world):
                full = O.point_add(cv, full, G[i])
            out["sum%d" % cv] = bool(r == rank and w == world and (fn(G[rank]) == full).all())
            out["identity%d" % cv] = bool((fn(np.zeros(8, dtype=np.uint64) if rank else G[0]) == G[0]).all())
            # the second collective (index-cyclic IPA): byte blocks come back in rank order
            blk = (np.arange(96, dtype=np.uint8) + 7 * rank).astype(np.uint8)
            got = np.asarray(fe.gather(blk)).reshape(world, 96)
            out["gather%d" % cv] = bool(all((got[r2] == (np.arange(96, dtype=np.uint8) + 7 * r2).astype(np.uint8)).all() for r2 in range(world)))
            P.enable_window_sharding(fe, cv, E.host_points_sum, 0, 1)
            out["off%d" % cv] = fe.args[1] == 1 and fe.args[2] is None and fe.gather is None
        dist.barrier()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_enable_window_sharding_gloo_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wshard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        assert all(out.values()), (rank, out)
