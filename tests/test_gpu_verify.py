"""GPU parity of r1cs::Verifier::verify (src/r1cs/verifier.rs:549-600) and batch_verify (:604-691): the engine
must accept / reject exactly like the CPU oracle on proofs from either side, including the reference's own
negative cases (tests/r1cs_secq256k1.rs:342-356, 395-411, 447-475)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = bytes([7]) * 32
OK, E_VERIFICATION, E_GENS, E_FORMAT = 0, -4, -5, -6


@pytest.fixture(scope="module", params=[0, 1], ids=["secq256k1", "zorro"])
def eng(request):
    import ark_bulletproofs_amd as A

    e = A.Engine(curve=request.param)
    e.gens_derive(128)
    yield e
    e.close()


CASES = [
    (0, [1]), (0, [2]), (0, [3]), (0, [7]), (0, [24]), (0, [42]),
    (2, [3, 4, 6, 1, 40, 9]), (2, [3, 4, 6, 1, 40, 10]),
    (1, [2, 3]), (1, [10, 1000]), (1, [63, (1 << 63) - 5]), (1, [10, 1024]), (1, [32, 1 << 32]),
    (3, [13, 0]), (3, [100, 0]), (3, [13, 1]), (4, [3, 8, 0]), (4, [3, 8, 1]),
]


@pytest.mark.parametrize("sc,prm", CASES)
def test_verify_matches_oracle(eng, oracle, sc, prm):
    O, cv = oracle, eng.curve
    ref = O.r1cs_prove(cv, sc, prm, SEED, 128, m_cap=128)
    assert ref.rc == 0
    rc_ref = O.r1cs_verify(cv, sc, prm, 128, ref.proof, ref.commitments, ref.publics)
    rc = eng.verify_scenario(sc, prm, ref.proof, ref.commitments, ref.publics)
    assert (rc == OK) == (rc_ref == 0)
    if rc_ref != 0:
        assert rc == E_VERIFICATION
    # a proof produced by the GPU prover verifies the same way
    got = eng.prove_scenario(sc, prm, SEED, m_cap=128)
    assert eng.verify_scenario(sc, prm, got.proof, got.commitments, got.publics) == rc


def test_verify_rejects_tampering(eng, oracle):
    O, cv = oracle, eng.curve
    sc, prm = 1, [16, 12345]
    ref = O.r1cs_prove(cv, sc, prm, SEED, 128)
    assert eng.verify_scenario(sc, prm, ref.proof, ref.commitments, ref.publics) == OK
    # flipped scalar byte (t_x): still well-formed, must fail the mega-check
    bad = bytearray(ref.proof)
    bad[11 * 33] ^= 1
    assert eng.verify_scenario(sc, prm, bytes(bad), ref.commitments, ref.publics) == E_VERIFICATION
    # malformed encodings -> FormatError, like R1CSProof::from_bytes
    assert eng.verify_scenario(sc, prm, ref.proof[:-1], ref.commitments, ref.publics) == E_FORMAT
    bad = bytearray(ref.proof)
    bad[32] = 0xC0
    assert eng.verify_scenario(sc, prm, bytes(bad), ref.commitments, ref.publics) == E_FORMAT
    # wrong commitment
    cm = ref.commitments.copy()
    cm[0] = O.generator(cv)
    assert eng.verify_scenario(sc, prm, ref.proof, cm, ref.publics) == E_VERIFICATION
    # identity point where the verifier validates (A_I1): rejected before the MSM
    bad = bytearray(ref.proof)
    bad[0:33] = bytes(32) + b"\x40"
    assert eng.verify_scenario(sc, prm, bytes(bad), ref.commitments, ref.publics) == E_VERIFICATION
    # statement larger than the installed generators
    big = O.r1cs_prove(cv, 3, [200, 0], SEED, 256, m_cap=8)
    assert eng.verify_scenario(3, [200, 0], big.proof, big.commitments, big.publics) == E_GENS


def test_batch_range_proof_gadget(eng, oracle):
    """tests/r1cs_secq256k1.rs:447-475: mixed sizes 16/32/64 in one batch, with the two negative sets"""
    O, cv = oracle, eng.curve

    def run(vals):
        inst = []
        for i, (v, n) in enumerate(vals):
            pr = O.r1cs_prove(cv, O.SC_RANGE, [n, v], bytes([9 + i]) * 32, 128)
            assert pr.rc == 0
            inst.append((O.SC_RANGE, [n, v], pr.proof, pr.commitments, pr.publics))
        rc_ref = O.batch_verify(cv, inst, 128, bytes([5]) * 32)
        rc, _ = eng.batch_verify(inst, bytes([5]) * 32)
        assert (rc == OK) == (rc_ref == 0)
        return rc

    assert run([(0, 16)]) == OK
    assert run([(0, 16), (3, 16), ((1 << 16) - 1, 16), (1 << 16, 32)]) == OK
    assert run([(0, 16), (3, 16), (1 << 16, 16), (1 << 16, 32)]) == E_VERIFICATION
    assert run([(0, 16), (3, 16), ((1 << 16) - 1, 16), (1 << 16, 32), (1 << 63, 64)]) == OK
    assert run([(0, 16), (3, 16), ((1 << 16) - 1, 16), (1 << 32, 32), (1 << 63, 64)]) == E_VERIFICATION


def test_batch_mixed_scenarios_and_many(eng, oracle):
    O, cv = oracle, eng.curve
    rc, _, pt = eng.batch_verify([], bytes([6]) * 32, want_point=True)   # empty batch: Ok, like the reference's MSM of nothing
    assert rc == OK and not pt.any()
    inst = []
    for i in range(12):
        sc, prm = [(0, [5]), (3, [20, 0]), (1, [8, 200]), (4, [2, 8, 0])][i % 4]
        pr = eng.prove_scenario(sc, prm, bytes([20 + i]) * 32, m_cap=32)
        inst.append((sc, prm, pr.proof, pr.commitments, pr.publics))
    rc, timing = eng.batch_verify(inst, bytes([6]) * 32)
    assert rc == OK and timing[0] > 0
    assert O.batch_verify(cv, inst, 128, bytes([6]) * 32) == 0
    # corrupt one proof in the middle: the whole batch must fail
    sc, prm, proof, cm, pb = inst[5]
    bad = bytearray(proof)
    bad[11 * 33 + 40] ^= 2
    inst[5] = (sc, prm, bytes(bad), cm, pb)
    rc, _ = eng.batch_verify(inst, bytes([6]) * 32)
    assert rc == E_VERIFICATION


def test_proof_sharded_batch_points_sum_to_identity(eng, oracle):
    """whole proofs per GPU; the per-shard mega-check points (with alpha_skip) sum to the identity
    exactly when the unsharded batch verifies — emulated here on one GPU with two shards."""
    from ark_bulletproofs_amd import engine as E
    from ark_bulletproofs_amd.parallel import shard_range

    O, cv = oracle, eng.curve
    inst = []
    for i in range(7):
        sc, prm = [(1, [16, 99 + i]), (3, [20, 0]), (0, [4])][i % 3]
        pr = eng.prove_scenario(sc, prm, bytes([40 + i]) * 32, m_cap=32)
        inst.append((sc, prm, pr.proof, pr.commitments, pr.publics))
    seed = bytes([8]) * 32
    rc_full, _, pt_full = eng.batch_verify(inst, seed, want_point=True)
    assert rc_full == OK and not pt_full.any()
    pts = []
    for r in range(2):
        lo, hi = shard_range(len(inst), r, 2)
        rc, _, pt = eng.batch_verify(inst[lo:hi], seed, alpha_skip=lo, want_point=True)
        assert rc == OK
        pts.append(pt)
    assert not E.host_points_sum(cv, np.stack(pts)).any()
    # one bad proof: its shard's point is not the identity, and neither is the sum
    sc, prm, proof, cm, pb = inst[5]
    bad = bytearray(proof)
    bad[11 * 33 + 40] ^= 2
    inst[5] = (sc, prm, bytes(bad), cm, pb)
    pts = []
    for r in range(2):
        lo, hi = shard_range(len(inst), r, 2)
        rc, _, pt = eng.batch_verify(inst[lo:hi], seed, alpha_skip=lo, want_point=True)
        pts.append(pt)
    assert E.host_points_sum(cv, np.stack(pts)).any()
    # the sharded points equal the unsharded mega-check value (same alphas by position)
    rc_full, _, pt_full = eng.batch_verify(inst, seed, want_point=True)
    assert rc_full == E_VERIFICATION
    assert (E.host_points_sum(cv, np.stack(pts)) == pt_full).all()


def test_batch_pipeline_many_blocks(eng, oracle):
    """batches longer than one pipeline block (512 instances): mixed templates inside and across blocks, two-phase (shuffle)
    statements in between, a failing instance in a late block, the first error in instance order when a format error and a
    verification error are both present, and the check point of the whole batch against two half batches (alpha_skip)."""
    from ark_bulletproofs_amd import engine as E

    O, cv = oracle, eng.curve
    kinds = [(3, [20, 0]), (1, [8, 200]), (4, [2, 8, 0]), (0, [5]), (3, [20, 0]), (3, [9, 0])]
    distinct = []
    for i, (sc, prm) in enumerate(kinds):
        pr = eng.prove_scenario(sc, prm, bytes([60 + i]) * 32, m_cap=32)
        distinct.append((sc, prm, pr.proof, pr.commitments, pr.publics))
    n = 1300
    inst = [distinct[(i * 7 + i // 100) % len(distinct)] for i in range(n)]
    seed = bytes([9]) * 32
    rc, _, pt = eng.batch_verify(inst, seed, want_point=True)
    assert rc == OK and not pt.any()
    # the oracle agrees on a prefix it can do in seconds (same alphas by position)
    assert O.batch_verify(cv, inst[:40], 128, seed) == 0
    # one verification failure in the third block
    sc, prm, proof, cm, pb = inst[1100]
    bad = bytearray(proof)
    bad[11 * 33 + 40] ^= 2   # t_x_blinding
    bad_inst = list(inst)
    bad_inst[1100] = (sc, prm, bytes(bad), cm, pb)
    rc, _, pt_bad = eng.batch_verify(bad_inst, seed, want_point=True)
    assert rc == E_VERIFICATION and pt_bad.any()
    # halves with alpha_skip reproduce the whole batch's check point
    lo = 650
    _, _, p0 = eng.batch_verify(bad_inst[:lo], seed, alpha_skip=0, want_point=True)
    _, _, p1 = eng.batch_verify(bad_inst[lo:], seed, alpha_skip=lo, want_point=True)
    assert (E.host_points_sum(cv, np.stack([p0, p1])) == pt_bad).all()
    # a malformed encoding late in the batch is a FormatError even though an earlier instance fails verification
    # (the reference decodes every proof before batch_verify runs)
    sc, prm, proof, cm, pb = inst[1200]
    worse = bytearray(proof)
    worse[32] |= 0x20        # reserved flag bit of the first compressed point
    bad_inst[1200] = (sc, prm, bytes(worse), cm, pb)
    rc, _ = eng.batch_verify(bad_inst, seed)
    assert rc == E_FORMAT
    # an x coordinate that is not on the curve, in the last block (its square root is taken by the decode that runs one block
    # ahead of the replay): FormatError as well, also when only an earlier instance fails verification
    sc, prm, proof, cm, pb = inst[1290]
    off_curve = None
    for delta in range(1, 40):
        cand = bytearray(proof)
        cand[0] = (cand[0] + delta) & 0xFF
        _, ok = eng.debug_decompress(bytes(cand[:33]))
        if not ok[0]:
            off_curve = bytes(cand)
            break
    assert off_curve is not None
    late = list(inst)
    late[1290] = (sc, prm, off_curve, cm, pb)
    rc, _ = eng.batch_verify(late, seed)
    assert rc == E_FORMAT
    sc1, prm1, proof1, cm1, pb1 = inst[1100]
    late[20] = bad_inst[1100] if inst[20][0] == sc1 and inst[20][1] == prm1 else late[20]
    late[1100] = (sc1, prm1, bytes(bad), cm1, pb1)
    rc, _ = eng.batch_verify(late, seed)
    assert rc == E_FORMAT
    # wrong commitment (statement mismatch) in the first block, everything else fine
    sc, prm, proof, cm, pb = inst[3]
    other = inst[4][3] if len(inst[4][3]) == len(cm) else cm[::-1].copy()
    mism = list(inst)
    mism[3] = (sc, prm, proof, other if (other != cm).any() else cm[::-1].copy(), pb)
    if (np.asarray(mism[3][3]) != np.asarray(cm)).any():
        rc, _ = eng.batch_verify(mism, seed)
        assert rc == E_VERIFICATION


def test_two_batches_in_flight_on_two_ctxs_with_divided_host_pools(eng, oracle):
    """bench.py keeps two batch_verify calls in flight, each on its own ctx (shared generator tables) with a host pool of its
    own size (BP_TUNE_HOST_THREADS): the same check points and decisions as one call at a time with the default pool."""
    import threading

    import ark_bulletproofs_amd as A

    cv = eng.curve
    kinds = [(3, [20, 0]), (1, [8, 200]), (0, [5])]
    distinct = []
    for i, (sc, prm) in enumerate(kinds):
        pr = eng.prove_scenario(sc, prm, bytes([90 + i]) * 32, m_cap=32)
        distinct.append((sc, prm, pr.proof, pr.commitments, pr.publics))
    good = [distinct[(i * 5 + i // 64) % len(distinct)] for i in range(700)]
    sc, prm, proof, cm, pb = good[650]
    tampered = bytearray(proof)
    tampered[11 * 33 + 40] ^= 4
    bad = list(good)
    bad[650] = (sc, prm, bytes(tampered), cm, pb)
    seed = bytes([4]) * 32
    rc0, _, p_good = eng.batch_verify(good, seed, want_point=True)
    rc1, _, p_bad = eng.batch_verify(bad, seed, want_point=True)
    assert rc0 == OK and rc1 == E_VERIFICATION and not p_good.any() and p_bad.any()
    with pytest.raises(Exception):
        eng.set_tuning(6, 100000)   # out of range
    e2 = A.Engine(curve=cv)
    e2.share_gens_from(eng)
    try:
        eng.set_tuning(6, 3)
        e2.set_tuning(6, 2)
        out = {}

        def run(tag, e, inst):
            out[tag] = [e.batch_verify(inst, seed, want_point=True) for _ in range(3)]

        th = [threading.Thread(target=run, args=("a", eng, good)), threading.Thread(target=run, args=("b", e2, bad))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for rc, _, pt in out["a"]:
            assert rc == OK and not pt.any()
        for rc, _, pt in out["b"]:
            assert rc == E_VERIFICATION and (pt == p_bad).all()
    finally:
        eng.set_tuning(6, 0)
        e2.close()


def test_verification_gh_scalars_match_reference_formulas(eng, oracle):
    """bp_r1cs_verification_gh against an independent big-integer evaluation of verifier.rs:465-514 / inner_product_proof.rs:279-311:
    s[i] = prod_j u_j^(+1 if bit (k-1-j)... ) in the reference's doubling order, g and h with the phase separator u on i >= n1,
    zero-padded wL/wR/wO beyond n."""
    O, cv = oracle, eng.curve
    FR = O.fid(cv, True)
    r = O.modulus(FR)
    import random

    rnd = random.Random(5)
    k, n, n1 = 6, 50, 20
    N = 1 << k
    val = lambda: rnd.randrange(1, r)   # noqa: E731
    y, x, u, a, b = val(), val(), val(), val(), val()
    ch = [val() for _ in range(k)]
    wL = [rnd.randrange(r) for _ in range(n)]
    wR = [rnd.randrange(r) for _ in range(n)]
    wO = [rnd.randrange(r) for _ in range(n)]
    wL[3] = wR[7] = wO[11] = 0
    inv = lambda v: pow(v, r - 2, r)   # noqa: E731
    # s vector exactly as the reference builds it (:302-311): s[0] = prod u_j^-1; s[i] = s[i - 2^lg] * u_sq[k-1-lg], lg = floor(log2 i)
    allinv = 1
    for c in ch:
        allinv = allinv * inv(c) % r
    u_sq = [c * c % r for c in ch]
    s = [allinv]
    for i in range(1, N):
        lg = i.bit_length() - 1
        s.append(s[i - (1 << lg)] * u_sq[k - 1 - lg] % r)
    y_inv = inv(y)
    g_exp, h_exp = [], []
    for i in range(N):
        wl = wL[i] if i < n else 0
        wr = wR[i] if i < n else 0
        wo = wO[i] if i < n else 0
        yni = pow(y_inv, i, r)
        uo1 = 1 if i < n1 else u
        g_exp.append(uo1 * (x * yni * wr - a * s[i]) % r)
        h_exp.append(uo1 * (yni * (x * wl + wo - b * s[N - 1 - i]) - 1) % r)
    M = lambda v: O.fe_from_int(FR, v)   # noqa: E731
    g, h = eng.verification_gh(n1, np.array([M(v) for v in wL]), np.array([M(v) for v in wR]), np.array([M(v) for v in wO]), M(y), M(x), M(u), M(a), M(b),
                               np.array([M(c) for c in ch]))
    got_g = [O.limbs_to_int(v) for v in g]
    got_h = [O.limbs_to_int(v) for v in h]
    assert got_g == g_exp and got_h == h_exp


def test_template_cache_eviction_keeps_what_the_block_uses(oracle):
    """ADVICE r02 (use-after-free in batch_verify_core): a long-lived ctx that has seen more circuit shapes than its template cache
    holds, then a block that mixes a NEW shape (first) with shapes whose templates were cached when the replay workers ran.  The
    eviction must not take a template an instance of the current block resolves to (its recording is already released)."""
    import ark_bulletproofs_amd as A

    O, cv = oracle, 0
    e = A.Engine(curve=cv)
    e.gens_derive(128)
    try:
        seed = bytes([11]) * 32

        def stmt(n, tag):
            pr = e.prove_scenario(3, [n, 0], bytes([tag & 255]) * 32, m_cap=8)   # square chain with n multipliers: one shape per n
            return (3, [n, 0], pr.proof, pr.commitments, pr.publics)

        shapes = {n: stmt(n, n) for n in range(2, 76)}
        # fill the cache past its bound: 70 shapes, each twice (a shape used once is recorded inside its own replay)
        first = [shapes[n] for n in range(2, 72)] * 2
        rc, _, pt = e.batch_verify(first, seed, want_point=True)
        assert rc == OK and not pt.any()
        # new shapes first, then shapes cached by the call above, all in one block; twice each so that every one has a shared source
        mixed = ([shapes[n] for n in (72, 73, 74, 75)] + [shapes[n] for n in range(40, 72)]) * 2
        rc, _, pt = e.batch_verify(mixed, seed, want_point=True)
        assert rc == OK and not pt.any()
        assert O.batch_verify(cv, mixed[:12], 128, seed) == 0
        # and a failing instance among them is still found
        sc, prm, proof, cm, pb = mixed[20]
        bad = bytearray(proof)
        bad[11 * 33 + 40] ^= 2
        mixed[20] = (sc, prm, bytes(bad), cm, pb)
        rc, _ = e.batch_verify(mixed, seed)
        assert rc == E_VERIFICATION
    finally:
        e.close()
