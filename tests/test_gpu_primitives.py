"""GPU parity tests, through the C ABI, of the kernels' field / group arithmetic and the MSM against the
CPU oracle on the same seeded inputs (bit-exact: integer work)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[0, 1], ids=["secq256k1", "zorro"])
def eng(request):
    import ark_bulletproofs_amd as A

    e = A.Engine(curve=request.param)
    yield e
    e.close()


def test_field_ops(eng, oracle):
    O = oracle
    n = 300
    for sf in (False, True):
        f = O.fid(eng.curve, sf)
        a = O.fe_rand(f, bytes([11]) * 32, n)
        b = O.fe_rand(f, bytes([12]) * 32, n)
        p = O.modulus(f)
        for i, v in enumerate([0, 1, p - 1, p - 2, 2, (1 << 255) % p]):
            a[i] = O.fe_from_int(f, v)
            b[i] = O.fe_from_int(f, [p - 1, 0, p - 1, 1, p - 2, 3][i])
        for op, name in [(0, "mul"), (1, "add"), (2, "sub")]:
            got = eng.debug_field_op(f, op, a, b)
            exp = np.array([O.fe_op(name, f, a[i], b[i]) for i in range(n)])
            assert (got == exp).all(), name
        got = eng.debug_field_op(f, 3, a, b)
        assert (got == np.array([O.fe_op("mul", f, a[i], a[i]) for i in range(n)])).all()
        got = eng.debug_field_op(f, 4, a[2:66], b[2:66])
        assert (got == np.array([O.fe_op("inv", f, a[i]) for i in range(2, 66)])).all()


def test_point_ops(eng, oracle):
    O = oracle
    cv = eng.curve
    n = 64
    G, H = O.bp_gens(cv, n)
    FR = O.fid(cv, True)
    Q = H.copy()
    Q[0] = G[0]                      # P + P
    Q[1, 4:] = O.fe_op("sub", O.fid(cv, False), O.fe_from_int(O.fid(cv, False), 0), G[1, 4:])
    Q[1, :4] = G[1, :4]              # P + (-P)
    Q[2] = 0                         # P + identity
    P = G.copy()
    P[3] = 0                         # identity + Q
    for op in (0, 1):
        got = eng.debug_point_op(op, P, Q)
        exp = np.array([O.point_add(cv, P[i], Q[i]) for i in range(n)])
        assert (got == exp).all()
    assert not got[1].any()
    got = eng.debug_point_op(2, P, Q)
    assert (got == np.array([O.point_add(cv, P[i], P[i]) for i in range(n)])).all()
    k = O.fe_rand(FR, bytes([5]) * 32, n)
    r = O.modulus(FR)
    kc = np.array([O.int_to_limbs(O.fe_to_int(FR, x)) for x in k])
    kc[0] = 0
    kc[1] = O.int_to_limbs(r - 1)
    kc[2] = O.int_to_limbs(1)
    got = eng.debug_point_op(3, G, Q, kc)
    exp = np.array([O.scalar_mul(cv, G[i], O.fe_from_int(FR, O.limbs_to_int(kc[i]))) for i in range(n)])
    assert (got == exp).all()


def _rand_scalars(O, cv, n, seed):
    return O.fe_rand(O.fid(cv, True), bytes([seed]) * 32, n)


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 1000])
def test_msm_small(eng, oracle, n):
    O, cv = oracle, eng.curve
    G, H = O.bp_gens(cv, max(n, 2))
    bases = np.concatenate([G, H])[:n]
    sc = _rand_scalars(O, cv, n, 2)
    assert (eng.msm(bases, sc) == O.msm(cv, bases, sc)).all()


def test_msm_skew_falls_back_from_fixed_shape_pipeline(eng, oracle):
    """The fixed-shape MSM pipeline (msm.cuh 7) serves spread scalars; a bucket above 256 entries or a full bin region raises its
    overflow flag and the general path redoes the MSM.  Both kinds of skew, and the spread case at the same size, against the oracle."""
    O, cv = oracle, eng.curve
    FR = O.fid(cv, True)
    n = 5000
    G, H = O.bp_gens(cv, n // 2)
    bases = np.concatenate([G, H])
    rnd = _rand_scalars(O, cv, n, 7)
    assert (eng.msm(bases, rnd) == O.msm(cv, bases, rnd)).all()
    # 0/1/2 witness-like scalars: whole windows in one or two buckets
    small = np.array([O.fe_from_int(FR, (i * 7 + 3) % 3) for i in range(n)])
    assert (eng.msm(bases, small) == O.msm(cv, bases, small)).all()
    # half spread, half one repeated full-size scalar: the bins hold, single buckets outgrow the reduce step
    mixed = rnd.copy()
    mixed[n // 2:] = rnd[11]
    assert (eng.msm(bases, mixed) == O.msm(cv, bases, mixed)).all()


def test_msm_edge_cases(eng, oracle):
    O, cv = oracle, eng.curve
    FR = O.fid(cv, True)
    n = 512
    G, H = O.bp_gens(cv, n)
    r = O.modulus(FR)
    zero, one, rm1 = O.fe_from_int(FR, 0), O.fe_from_int(FR, 1), O.fe_from_int(FR, r - 1)
    # all-zero scalars -> identity
    assert not eng.msm(G, np.tile(zero, (n, 1))).any()
    # scalar = 1 everywhere: plain sum (one giant bucket per window: the skew path)
    ones = np.tile(one, (n, 1))
    assert (eng.msm(G, ones) == O.msm(cv, G, ones)).all()
    # scalar = r - 1 everywhere
    m1 = np.tile(rm1, (n, 1))
    assert (eng.msm(G, m1) == O.msm(cv, G, m1)).all()
    # duplicated bases, identity bases, P and -P with equal scalars (cancellation inside a bucket)
    sc = _rand_scalars(O, cv, n, 3)
    B = G.copy()
    B[1] = B[0]
    B[5] = 0
    B[7, :4] = B[6, :4]
    B[7, 4:] = O.fe_op("sub", O.fid(cv, False), O.fe_from_int(O.fid(cv, False), 0), B[6, 4:])
    sc[7] = sc[6]
    sc[1] = sc[0]
    assert (eng.msm(B, sc) == O.msm(cv, B, sc)).all()
    # all bases equal and all scalars equal (every add in a bucket is a doubling)
    B2 = np.tile(G[3], (64, 1))
    s2 = np.tile(sc[9], (64, 1))
    assert (eng.msm(B2, s2) == O.msm(cv, B2, s2)).all()
    # canonical-integer scalars flag
    can = np.array([O.int_to_limbs(O.fe_to_int(FR, x)) for x in sc])
    assert (eng.msm(G, can, canonical=True) == O.msm(cv, G, sc)).all()
    # small 0/1 witness-like scalars
    bits = np.array([O.fe_from_int(FR, (i * 7 + 3) % 2) for i in range(n)])
    assert (eng.msm(H, bits) == O.msm(cv, H, bits)).all()
    with pytest.raises(ValueError):
        eng.msm(G[:4], sc[:3])


def test_msm_two_level_sort_forced_small(eng, oracle):
    """two-level sort with a handful of terms (bins of a few entries, empty bins, the narrow top window in LDS)"""
    O, cv = oracle, eng.curve
    eng.set_tuning(1, 1)
    eng.set_tuning(3, 1)      # running-sum window aggregation for every bucket count
    try:
        for n in (1, 2, 31, 33, 257, 1000):
            G, H = O.bp_gens(cv, max(n, 2))
            bases = np.concatenate([G, H])[:n]
            sc = _rand_scalars(O, cv, n, (40 + n) % 200)
            assert (eng.msm(bases, sc) == O.msm(cv, bases, sc)).all()
    finally:
        eng.set_tuning(1, 64)
        eng.set_tuning(3, 1 << 18)


def test_msm_two_level_sort_sizes(eng, oracle):
    """sizes that take the two-level (binned) sort: uniform scalars, skew that overflows a bin region (falls back to the slot /
    exact paths), 0/1 and r-1 vectors, zero scalars and identity bases in between, canonical input"""
    O, cv = oracle, eng.curve
    FR = O.fid(cv, True)
    r = O.modulus(FR)
    n = 40000
    G, H = O.bp_gens(cv, n // 2)
    bases = np.concatenate([G, H])
    sc = _rand_scalars(O, cv, n, 12)
    bases[17] = 0
    sc[100:200] = 0
    assert (eng.msm(bases, sc) == O.msm(cv, bases, sc)).all()
    can = np.array([O.int_to_limbs(O.fe_to_int(FR, x)) for x in sc[:5000]])
    assert (eng.msm(bases[:5000], can, canonical=True) == O.msm(cv, bases[:5000], sc[:5000])).all()
    skew = sc.copy()
    skew[: n // 2] = sc[3]                       # half the terms share one scalar: their bins overflow
    assert (eng.msm(bases, skew) == O.msm(cv, bases, skew)).all()
    bits = np.array([O.fe_from_int(FR, (i * 5 + 1) % 2) for i in range(n)])
    assert (eng.msm(bases, bits) == O.msm(cv, bases, bits)).all()
    m1 = np.tile(O.fe_from_int(FR, r - 1), (n, 1))
    assert (eng.msm(bases, m1) == O.msm(cv, bases, m1)).all()
    small = np.array([O.fe_from_int(FR, (i * 2654435761) % (1 << 20)) for i in range(n)])   # only the low windows populated
    assert (eng.msm(bases, small) == O.msm(cv, bases, small)).all()
    for m in (4096, 4097, 8191, 8193, 16385):
        assert (eng.msm(bases[:m], sc[:m]) == O.msm(cv, bases[:m], sc[:m])).all()


def test_msm_cfg2_2pow16(eng, oracle):
    """BASELINE.json configs[1]: 2^16-term MSM, bases = BulletproofGens(2^15) G||H, scalars from ChaCha20 seed [2;32]"""
    O, cv = oracle, eng.curve
    n = 1 << 16
    G, H = O.bp_gens(cv, n // 2)
    bases = np.concatenate([G, H])
    sc = _rand_scalars(O, cv, n, 2)
    got = eng.msm(bases, sc)
    assert (got == O.msm(cv, bases, sc)).all()
    # device-resident path + linearity: msm(b, s) + msm(b, s') == msm(b, s + s')
    db, ds = eng.upload_points(bases), eng.upload_scalars(sc)
    assert (eng.msm_dev(db, ds, n) == got).all()
    db.free()
    ds.free()


def test_msm_window_sharding(eng, oracle):
    """north_star's multi-GPU MSM: GPU g accumulates a range of Pippenger windows; the partials (already weighted by
    2^(c*w)) must add up to the full MSM.  Emulated on one GPU with 1, 3 and 8 'ranks'."""
    from ark_bulletproofs_amd import engine as E
    from ark_bulletproofs_amd.parallel import shard_range

    O, cv = oracle, eng.curve
    n = 5000
    G, H = O.bp_gens(cv, n // 2)
    bases = np.concatenate([G, H])
    sc = _rand_scalars(O, cv, n, 6)
    exp = O.msm(cv, bases, sc)
    db, ds = eng.upload_points(bases), eng.upload_scalars(sc)
    W, c = E.msm_window_count(cv, n)
    assert W * c >= 255 and W > 8
    for world in (1, 3, 8):
        parts = []
        for r in range(world):
            lo, hi = shard_range(W, r, world)
            parts.append(eng.msm_dev_windows(db, ds, n, lo, hi))
        assert (E.host_points_sum(cv, np.stack(parts)) == exp).all()
    assert not eng.msm_dev_windows(db, ds, n, 3, 3).any()          # empty range -> identity
    assert (eng.msm_dev_windows(db, ds, n, 0, W) == exp).all()
    db.free()
    ds.free()


def test_point_decompression_on_gpu(eng, oracle):
    """the wire-codec square roots (R1CSProof::from_bytes) as a kernel: every encoding ark accepts decodes to the same point,
    everything it rejects is rejected"""
    O, cv = oracle, eng.curve
    G, H = O.bp_gens(cv, 200)
    pts = np.concatenate([G, H])
    enc = b"".join(O.point_ser(cv, p, True) for p in pts) + bytes(32) + b"\x40"
    out, ok = eng.debug_decompress(enc)
    assert ok.all() and (out[:-1] == pts).all() and not out[-1].any()
    # flipped sign flag -> the other root
    flipped = bytearray(enc[:33])
    flipped[32] ^= 0x80
    o2, ok2 = eng.debug_decompress(bytes(flipped))
    assert ok2[0] and (o2[0, :4] == pts[0, :4]).all() and (O.point_add(cv, o2[0], pts[0]) == 0).all()
    # x values that are not on the curve, bad flags, x >= p, identity flag with x != 0
    bad = []
    x = 1
    q = O.modulus(O.fid(cv, False))
    while len(bad) < 5:
        x += 1
        b = x.to_bytes(32, "little") + b"\x00"
        if O.point_deser_compressed(cv, b) is None:
            bad.append(b)
    bad += [enc[:32] + b"\xc0", enc[:32] + b"\x01", q.to_bytes(32, "little") + b"\x00", enc[:32] + b"\x40"]
    o3, ok3 = eng.debug_decompress(b"".join(bad))
    assert not ok3.any() and not o3.any()
    for b in bad:
        assert O.point_deser_compressed(cv, b) is None


def test_pedersen_commit_batch(eng, oracle):
    """PedersenGens::commit (src/generators.rs:39-44) for a batch, by fixed-base tables on the GPU, against the oracle's
    double-and-add; edge rows: zero value, zero blinding, both zero (identity), r-1, small values, single-window digits."""
    O = oracle
    cv = eng.curve
    FR = O.fid(cv, True)
    r = O.modulus(FR)
    n = 200
    v = O.fe_rand(FR, bytes([21]) * 32, n)
    b = O.fe_rand(FR, bytes([22]) * 32, n)
    edge = [(0, 5), (5, 0), (0, 0), (r - 1, r - 1), (1, 1), (255, 256), (1 << 248, 1 << 255 if (1 << 255) < r else 1 << 254), (r - 1, 1), (2**64 - 1, r - 2)]
    for i, (x, y) in enumerate(edge):
        v[i] = O.fe_from_int(FR, x)
        b[i] = O.fe_from_int(FR, y)
    got = eng.pedersen_commit_batch(v, b)
    exp = np.array([O.pedersen_commit(cv, v[i], b[i]) for i in range(n)])
    assert (got == exp).all()
    assert (got[2] == 0).all()
    assert eng.pedersen_commit_batch(v[:0], b[:0]).shape == (0, 8)
    # v*B - v*B: (v, 0) + (r - v, 0) style cancellation inside one commitment cannot occur; check the in-table doubling case
    # d*2^(8w)*B added onto the same point: value 2^8 + ... is covered by the random rows; here two equal windows of B and B_blinding
    one = np.array([O.fe_from_int(FR, 77)])
    assert (eng.pedersen_commit_batch(one, one)[0] == O.pedersen_commit(cv, one[0], one[0])).all()


def test_statement_built_on_gpu_equals_host_statement(eng):
    """bp_stmt_prover_create_dev (commitments as one GPU batch) must describe the same statement as the host-only constructor"""
    from ark_bulletproofs_amd import engine as E

    for sc, prm in [(0, [9]), (4, [5, 16, 0]), (2, [3, 4, 6, 1, 40, 9]), (3, [20, 0])]:
        seed = bytes([5, sc]) + bytes(30)
        a = E.Statement(eng.curve, sc, prm, seed)
        b = E.Statement(eng.curve, sc, prm, seed, engine=eng)
        ia, ib = a.info(), b.info()
        assert (ia[0] == ib[0]).all() and (ia[1] == ib[1]).all() and ia[2:] == ib[2:]
        a.free()
        b.free()


def test_msm_over_resident_generator_tables(oracle):
    """bp_msm_gens: the prover's msm call sites over BulletproofGens without re-uploading bases, e.g.
    A_I = msm([B_blinding] ++ G[..n] ++ H[..n]) (src/r1cs/prover.rs:516-559)"""
    import ark_bulletproofs_amd as A

    O = oracle
    for cv in (0, 1):
        e = A.Engine(curve=cv)
        e.gens_derive(300)
        G, H = O.bp_gens(cv, 300)
        _, Bb = O.pedersen_default(cv)
        n = 257
        sc = _rand_scalars(O, cv, 2 * n + 1, 77)
        got = e.msm_gens(n, sc, extra_bases=Bb.reshape(1, 8))
        assert (got == O.msm(cv, np.concatenate([G[:n], H[:n], Bb.reshape(1, 8)]), sc)).all()
        # G only with an offset (phase-2 slices G[n1..]), H only, extras only, nothing
        assert (e.msm_gens(40, sc[:40], use_H=False, off=100) == O.msm(cv, G[100:140], sc[:40])).all()
        assert (e.msm_gens(40, sc[:41], use_G=False, off=260, extra_bases=G[:1]) == O.msm(cv, np.concatenate([H[260:300], G[:1]]), sc[:41])).all()
        assert (e.msm_gens(0, sc[:3], extra_bases=H[5:8]) == O.msm(cv, H[5:8], sc[:3])).all()
        assert not e.msm_gens(0, sc[:0]).any()
        with pytest.raises(A.ArkbpError) as ei:
            e.msm_gens(41, sc[:82], off=260)
        assert ei.value.code == -5
        e.close()


def test_reference_held_constants_on_gpu(eng, oracle):
    """exp_iter(2) -> 1, 2, 4, 8 (src/util.rs:147-157) and inner_product = 40 (src/util.rs:160-166, src/inner_product_proof.rs:
    556-562) through the kernels' own power-table and inner-product code; plus longer runs against the oracle"""
    O = oracle
    e, curve = eng, eng.curve
    fr = O.fid(curve, True)
    two = O.fe_from_int(fr, 2)
    got = e.debug_exp_iter(two, 4)
    for i, want in enumerate([1, 2, 4, 8]):
        assert (got[i] == O.fe_from_int(fr, want)).all()
    x = O.fe_rand(fr, bytes([8]) * 32, 1)[0]
    assert (e.debug_exp_iter(x, 1000) == O.exp_iter(fr, x, 1000)).all()
    a = [O.fe_from_int(fr, v) for v in (1, 2, 3, 4)]
    b = [O.fe_from_int(fr, v) for v in (2, 3, 4, 5)]
    assert (e.debug_inner_product(a, b) == O.fe_from_int(fr, 40)).all()
    va, vb = O.fe_rand(fr, bytes([9]) * 32, 777), O.fe_rand(fr, bytes([10]) * 32, 777)
    assert (e.debug_inner_product(va, vb) == O.inner_product(fr, va, vb)).all()
    # secp256k1's Fr (= secq256k1's base field, the field of the reference's second inner_product test): GPU field ops
    if curve == 0:
        fq = O.fid(0, False)
        A = np.array([O.fe_from_int(fq, v) for v in (1, 2, 3, 4)])
        B = np.array([O.fe_from_int(fq, v) for v in (2, 3, 4, 5)])
        prod = e.debug_field_op(fq, 0, A, B)
        s = e.debug_field_op(fq, 1, e.debug_field_op(fq, 1, prod[0:1], prod[1:2]), e.debug_field_op(fq, 1, prod[2:3], prod[3:4]))
        assert (s[0] == O.fe_from_int(fq, 40)).all()


def test_msm_gens_fixed_base_rows_match_ordinary_schedule(eng, oracle):
    """bp_msm_gens over the resident tables through the fixed-base rows (bp_gens_msm_tables) and through the ordinary schedule:
    same point; against the oracle's msm at 2 x 3000 terms; skewed scalars (0/1) fall back and still agree"""
    O = oracle
    cv = eng.curve
    n = 3000
    eng.gens_derive(4096)
    Go, Ho = O.bp_gens(cv, n)
    fr = O.fid(cv, True)
    sc = O.fe_rand(fr, bytes([21]) * 32, 2 * n)
    want = O.msm(cv, np.concatenate([Go, Ho]), sc)
    assert (eng.msm_gens(n, sc) == want).all()
    eng.gens_msm_tables(4096)
    eng.set_tuning(5, 4096)
    assert (eng.msm_gens(n, sc) == want).all()
    assert (eng.msm_gens(n, sc[:n], use_H=False) == O.msm(cv, Go, sc[:n])).all()
    bits = np.array([O.fe_from_int(fr, (i * 7) & 1) for i in range(2 * n)])
    assert (eng.msm_gens(n, bits) == O.msm(cv, np.concatenate([Go, Ho]), bits)).all()
    eng.gens_msm_tables(0)


def test_native_rccl_collectives_world_of_one():
    """bp_rccl_unique_id / bp_ctx_rccl_init / the library's own ncclAllGather (include/arkbp.h "Native collectives") on the one GPU of
    the test box: the communicator comes up (RCCL bound at run time), an all-gather returns the rank's block, the stats count it,
    and a ctx with a world-1 communicator still computes the same MSM (nothing to exchange)."""
    import ark_bulletproofs_amd as A
    from ark_bulletproofs_amd import engine as E
    from oracle import pyoracle as O

    e = A.Engine(curve=0)
    try:
        uid = E.rccl_unique_id()
        assert len(uid) == 128 and any(uid)
        e.rccl_init(uid, 0, 1)
        blk = np.arange(96, dtype=np.uint8)
        out = e.debug_rccl_allgather(blk, 1)
        assert out.shape == (1, 96) and (out[0] == blk).all()
        big = (np.arange(1 << 16, dtype=np.uint32) * 2654435761 >> 7).astype(np.uint8)
        assert (e.debug_rccl_allgather(big, 1)[0] == big).all()
        n, secs = e.collective_stats()
        assert n == 2 and secs > 0
        G, H = O.bp_gens(0, 64)
        bases = np.concatenate([G, H])
        sc = O.fe_rand(O.fid(0, True), bytes([12]) * 32, 128)
        assert (e.msm(bases, sc) == O.msm(0, bases, sc)).all()
        e.rccl_shutdown()
        with pytest.raises(E.ArkbpError):
            e.debug_rccl_allgather(blk, 1)
    finally:
        e.close()


def test_msm_glv_split_schedule_matches_ordinary_and_oracle(oracle):
    """secq256k1's mid-size MSMs split their scalars with the endomorphism (msm.cuh glv_split: s*P = k1*P + k2*phi(P), two
    128-bit halves): same value as the ordinary schedule (BP_TUNE_MSM_GLV_MIN turns the split off) and as the oracle, for spread
    scalars and for the scalars where the decomposition is extreme: 0, 1, r-1, (r-1)/2, lambda and lambda^2 (the halves swap roles),
    2^128 - 1, 2^128, 2^255, duplicates and identity bases, skew that overflows (ordinary schedule takes over)."""
    import ark_bulletproofs_amd as A

    O, cv = oracle, 0
    FR = O.fid(cv, True)
    r = O.modulus(FR)
    e = A.Engine(curve=cv)
    try:
        lam = next(l for l in (pow(g, (r - 1) // 3, r) for g in range(2, 20)) if l != 1)
        special = [0, 1, 2, r - 1, r - 2, (r - 1) // 2, (r + 1) // 2, lam, lam * lam % r, r - lam, (1 << 128) - 1, 1 << 128, (1 << 128) + 1, 1 << 255, (1 << 127), r // 3, 3]
        for n in (256, 1000, 5000, 1 << 15):
            G, H = O.bp_gens(cv, n // 2 + 1)
            bases = np.concatenate([G, H])[:n]
            sc = O.fe_rand(FR, bytes([n % 251]) * 32, n)
            for j, v in enumerate(special):
                sc[(j * 13) % n] = O.fe_from_int(FR, v)
            bases[5] = bases[4]
            bases[9] = 0
            e.set_tuning(7, 256)
            got_glv = e.msm(bases, sc)
            e.set_tuning(7, 1 << 40)
            got_plain = e.msm(bases, sc)
            assert (got_glv == got_plain).all(), n
            if n <= 5000:
                assert (got_glv == O.msm(cv, bases, sc)).all(), n
        # the special scalars alone, repeated: every half-term of a window in a handful of buckets
        n = 1024
        G, H = O.bp_gens(cv, n)
        sc = np.array([O.fe_from_int(FR, special[i % len(special)]) for i in range(n)])
        e.set_tuning(7, 256)
        assert (e.msm(G, sc) == O.msm(cv, G, sc)).all()
        # lane-per-addition trees (ARKBP_MSM_NOQUAD has no run-time switch): the quad-cooperative trees are what ran above; the same
        # inputs through the general path (two-level sort minimum above n) must agree
        sc = O.fe_rand(FR, bytes([77]) * 32, n)
        a = e.msm(G, sc)
        e.set_tuning(1, 1 << 30)
        e.set_tuning(7, 1 << 40)
        b = e.msm(G, sc)
        e.set_tuning(1, 64)
        assert (a == b).all() and (a == O.msm(cv, G, sc)).all()
    finally:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cv", [0, 1])
def test_msm_chunk_sizes_of_the_fixed_shape_pipeline(oracle, cv):
    """BP_TUNE_MSM_CHUNK_CAP: the mid-size MSM pipeline cuts every bucket into ceil(population / cap) chunks that share its entries
    evenly; any cap from 8 to 64 must give the same point (a sum of the same group elements) — also with a few heavy buckets (repeated
    scalars: up to 32 partials per bucket reach the one-step reduction), and at both ends of the range the knob accepts."""
    import ark_bulletproofs_amd as A

    O = oracle
    FR = O.fid(cv, True)
    e = A.Engine(curve=cv)
    try:
        for n in (700, 6000):
            G, H = O.bp_gens(cv, n // 2 + 1)
            bases = np.concatenate([G, H])[:n]
            sc = O.fe_rand(FR, bytes([n % 199, cv]) * 16, n)
            sc[10:150] = sc[3]               # one scalar 140 times: the same bucket of every window holds 140 of its entries
            sc[200:230] = O.fe_from_int(FR, 1)
            want = O.msm(cv, bases, sc)
            for cap in (0, 8, 9, 11, 12, 13, 16, 21, 32, 64):
                e.set_tuning(9, cap)
                assert (e.msm(bases, sc) == want).all(), (n, cap)
        for bad in (1, 7, 65):
            with pytest.raises(Exception):
                e.set_tuning(9, bad)
    finally:
        e.close()
