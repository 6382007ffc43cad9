"""Full-size GPU tests through size-independent properties (the CPU oracle cannot reach these sizes in seconds):
prove -> verify round trips at 2^14 / 2^16 constraints with tamper rejection, MSM linearity at 2^20 terms, batch verify
of mixed large proofs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import ark_bulletproofs_amd as A

    e = A.Engine(curve=0)
    e.gens_derive(1 << 16)
    yield e
    e.close()


def test_prove_verify_roundtrip_2pow16(eng):
    N = 1 << 16
    pr = eng.prove_scenario(3, [N, 0], bytes([3]) * 32, m_cap=8)
    assert len(pr.proof) == 539 + 66 * 16
    assert eng.verify_scenario(3, [N, 0], pr.proof, pr.commitments, pr.publics) == 0
    bad = bytearray(pr.proof)
    bad[-1] ^= 1  # ipp b
    assert eng.verify_scenario(3, [N, 0], bytes(bad), pr.commitments, pr.publics) in (-4, -6)
    wrong = pr.publics.copy()
    wrong[0, 0] ^= np.uint64(2)
    assert eng.verify_scenario(3, [N, 0], pr.proof, pr.commitments, wrong) == -4
    # a wrong witness (public output off by one) yields a proof that must not verify
    pr_bad = eng.prove_scenario(3, [N, 1], bytes([3]) * 32, m_cap=8)
    assert eng.verify_scenario(3, [N, 1], pr_bad.proof, pr_bad.commitments, pr_bad.publics) == -4


def test_cfg4_shape_prove_and_batch_verify(eng):
    """BASELINE cfg4 statement shape: 256 x 64-bit range proofs in one circuit (n = 2^14, m = 256, 0/1 witness vectors:
    the skewed-bucket path of the MSM), batch-verified together with a shuffle and a square chain of other sizes."""
    inst = []
    for i in range(3):
        pr = eng.prove_scenario(4, [256, 64, 0], bytes([4, i] + [4] * 30), m_cap=264)
        assert len(pr.commitments) == 256
        inst.append((4, [256, 64, 0], pr.proof, pr.commitments, pr.publics))
    pr = eng.prove_scenario(0, [1000], bytes([9]) * 32, m_cap=2008)       # k-shuffle: 2-phase, 1998 multipliers -> 2048
    inst.append((0, [1000], pr.proof, pr.commitments, pr.publics))
    pr = eng.prove_scenario(3, [5000, 0], bytes([8]) * 32, m_cap=8)         # non-power-of-two: padding path
    inst.append((3, [5000, 0], pr.proof, pr.commitments, pr.publics))
    for sc, prm, proof, cm, pb in inst:
        assert eng.verify_scenario(sc, prm, proof, cm, pb) == 0
    rc, _ = eng.batch_verify(inst, bytes([5]) * 32)
    assert rc == 0
    sc, prm, proof, cm, pb = inst[1]
    bad = bytearray(proof)
    bad[11 * 33 + 3] ^= 4
    inst[1] = (sc, prm, bytes(bad), cm, pb)
    rc, _ = eng.batch_verify(inst, bytes([5]) * 32)
    assert rc == -4


def test_msm_linearity_2pow20(eng):
    """msm(b || b, s1 || s2) == msm(b, s1) + msm(b, s2) and msm is invariant under a permutation of its terms"""
    from ark_bulletproofs_amd import engine as E

    n = 1 << 19
    G, H = eng.gens_download(1 << 16)
    bases = np.tile(np.concatenate([G, H]), (n // (1 << 17), 1))            # 2^19 bases (repeats are fine)
    rng = np.random.default_rng(7)
    s1 = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    s2 = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    s1[:, 3] >>= np.uint64(2)
    s2[:, 3] >>= np.uint64(2)
    db = eng.upload_points(np.concatenate([bases, bases]))
    ds = eng.upload_scalars(np.concatenate([s1, s2]))
    full = eng.msm_dev(db, ds, 2 * n)
    db1, ds1, ds2 = eng.upload_points(bases), eng.upload_scalars(s1), eng.upload_scalars(s2)
    a, b = eng.msm_dev(db1, ds1, n), eng.msm_dev(db1, ds2, n)
    assert (E.host_points_sum(0, np.stack([a, b])) == full).all()
    perm = rng.permutation(n)
    dbp, dsp = eng.upload_points(bases[perm]), eng.upload_scalars(s1[perm])
    assert (eng.msm_dev(dbp, dsp, n) == a).all()
    for d in (db, ds, db1, ds1, ds2, dbp, dsp):
        d.free()


def test_gpu_generator_derivation_matches_oracle_at_depth(eng, oracle):
    """BulletproofGens::new on the GPU (device Tonelli-Shanks over the recorded ChaCha20 attempts) against the ORACLE's
    GeneratorsChain (src/generators.rs:71-121, restated sequentially) deep into the chain (all indices up to 2^16), and
    against the product's own host derivation"""
    from ark_bulletproofs_amd import engine as E

    G, H = eng.gens_download(1 << 16)
    Go, Ho = oracle.bp_gens(0, 1 << 16)
    assert (G == Go).all() and (H == Ho).all()
    Gh = E.host_derive_generators(0, 0, 0, 1 << 16)
    Hh = E.host_derive_generators(0, 1, 0, 1 << 16)
    assert (G == Gh).all() and (H == Hh).all()


def test_cfg4_full_batch_4096_and_failing_check_point(eng, oracle):
    """BASELINE cfg4 at its real workload: batch_verify of 4096 proofs of 2^14 constraints (256 x 64-bit range proofs, m = 256).
    Accept; one corrupted proof -> VerificationError.  The oracle replays a 48-instance prefix containing the corrupted one with
    the same alphas: by linearity (valid instances contribute the identity) the FULL failing batch's mega-check point on the GPU
    must equal the oracle's point for the prefix."""
    from ark_bulletproofs_amd import engine as E

    O = oracle
    distinct = []
    for i in range(8):
        pr = eng.prove_scenario(E.SC_MULTI_RANGE, [256, 64, 0], bytes([4, i, 1] + [4] * 29), m_cap=264)
        distinct.append((E.SC_MULTI_RANGE, [256, 64, 0], pr.proof, pr.commitments, pr.publics))
    inst = [distinct[i % 8] for i in range(4096)]
    seed = bytes([5]) * 32
    rc, _, pt = eng.batch_verify(inst, seed, want_point=True)
    assert rc == 0 and not pt.any()
    bad_at = 37
    sc, prm, proof, cm, pb = inst[bad_at]
    bad = bytearray(proof)
    bad[11 * 33 + 40] ^= 8          # t_x_blinding
    inst[bad_at] = (sc, prm, bytes(bad), cm, pb)
    rc, _, pt = eng.batch_verify(inst, seed, want_point=True)
    assert rc == -4 and pt.any()
    orc, opt = O.batch_verify_point(0, inst[:48], 1 << 14, seed)
    assert orc == O.E_VERIFICATION
    assert (pt == opt).all(), "mega-check point of the failing full batch differs from the oracle's"
    # a second corrupted instance far behind the prefix changes the point
    sc, prm, proof, cm, pb = inst[3000]
    bad = bytearray(proof)
    bad[-1] ^= 1
    inst[3000] = (sc, prm, bytes(bad), cm, pb)
    rc, _, pt2 = eng.batch_verify(inst, seed, want_point=True)
    assert rc in (-4, -6)
    if rc == -4:
        assert (pt2 != pt).any()


@pytest.mark.parametrize("curve", [0, 1], ids=["secq256k1", "zorro"])
def test_cfg5_prove_verify_2pow22(curve):
    """BASELINE cfg5's size on one GPU: a 2^22-constraint square-chain proof (generator tables 2 x 256 MiB resident), prove ->
    verify -> tamper, on secq256k1 and on zorro"""
    import ark_bulletproofs_amd as A
    from ark_bulletproofs_amd import engine as E

    N = 1 << 22
    e = A.Engine(curve=curve)
    try:
        e.gens_derive(N)
        st = E.Statement(curve, E.SC_SQUARE_CHAIN, [N, 0], bytes([3 + curve]) * 32)
        commits, pubs, nm, nq = st.info(m_cap=8)
        assert nm == N and nq == 2 * N + 1
        proof, _ = st.prove(e)
        st.free()
        assert len(proof) == 539 + 66 * 22
        assert e.verify_scenario(E.SC_SQUARE_CHAIN, [N, 0], proof, commits, pubs) == 0
        bad = bytearray(proof)
        bad[2 * 33 + 7] ^= 1          # S1's x coordinate: off the curve (FormatError) or another point (VerificationError)
        assert e.verify_scenario(E.SC_SQUARE_CHAIN, [N, 0], bytes(bad), commits, pubs) in (-4, -6)
        bad = bytearray(proof)
        bad[-33] ^= 1                 # ipp a
        assert e.verify_scenario(E.SC_SQUARE_CHAIN, [N, 0], bytes(bad), commits, pubs) == -4
        wrong = pubs.copy()
        wrong[0, 1] ^= np.uint64(1)
        assert e.verify_scenario(E.SC_SQUARE_CHAIN, [N, 0], proof, commits, wrong) == -4
    finally:
        e.close()


def test_cfg5_2pow22_partitioned_across_four_ranks_equals_the_one_rank_proof():
    """BASELINE cfg5 AS A WHOLE on one GPU: the 2^22-constraint proof through north_star's partition — Pippenger windows of the
    commitment MSMs and the index-cyclic inner-product argument (2^20 elements per rank, gather at the frozen-tail length) — by FOUR
    ranks (threads, one Engine each, sharing ONE resident set of generator tables, fold tables and fixed-base rows), exchanging partial
    points and the tail vectors through an in-process all-gather.  Every rank must emit exactly the bytes of the one-rank proof,
    the partitioned verifier (window-sharded mega-check) must accept it, and the ranks must have kept the single-GPU MSM algorithm:
    fixed-base MSMs over their blocks of the commitments' terms and over the strided rows of the first round (four ranks: 2^21 + 1
    terms per rank — the fixed-base schedule's entry words hold up to 2^22 terms).  (the partition had run at 2^16, the
    size on one rank — never both; and the sharded prover stepped back to the ordinary schedule.)"""
    import threading

    import ark_bulletproofs_amd as A
    from ark_bulletproofs_amd import engine as E
    from ark_bulletproofs_amd import parallel as P

    N, world, cv = 1 << 22, 4, 0
    seed = bytes([9]) * 32
    single = A.Engine(curve=cv)
    try:
        single.gens_derive(N)
        st = E.Statement(cv, E.SC_SQUARE_CHAIN, [N, 0], seed)
        commits, pubs, nm, _ = st.info(m_cap=8)
        assert nm == N
        ref, _ = st.prove(single)
        st.free()
        assert single.verify_scenario(E.SC_SQUARE_CHAIN, [N, 0], ref, commits, pubs) == 0
        single.gens_fold_tables(N // 2, window_bits=4)          # the ranks index the same tables with a stride
        single.gens_msm_tables(N)                               # fixed-base rows: per-rank blocks of the commitments, strided rows in round 1
        bar = threading.Barrier(world)
        slots, out, errors = [None] * world, [None] * world, []
        gathers = [[0, 0] for _ in range(world)]                  # [64-byte point reduces, vector gathers]

        def run(rank):
            try:
                e = A.Engine(curve=cv)
                e.share_gens_from(single)

                def allgather(arr):
                    gathers[rank][1 if np.asarray(arr).size > 8 else 0] += 1
                    slots[rank] = np.array(arr, copy=True)
                    bar.wait()
                    res = np.stack(slots)
                    bar.wait()
                    return res

                P.enable_window_sharding(e, cv, E.host_points_sum, rank, world, allgather=allgather)
                s2 = E.Statement(cv, E.SC_SQUARE_CHAIN, [N, 0], seed)
                proof, _ = s2.prove(e)
                s2.free()
                n_prove = list(gathers[rank])
                fb = e.msm_stats()[1]
                rc = e.verify_scenario(E.SC_SQUARE_CHAIN, [N, 0], proof, commits, pubs)
                out[rank] = (proof, rc, n_prove, fb)
                P.enable_window_sharding(e, cv, E.host_points_sum, 0, 1)
                e.close()
            except Exception as ex:   # noqa: BLE001
                errors.append(ex)
                bar.abort()

        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        for r in range(world):
            proof, rc, (n_pts, n_vec), fb = out[r]
            assert fb >= 5, "the sharded prover stepped back from the fixed-base MSM schedule (%d runs on rank %d)" % (fb, r)
            assert proof == ref, "rank %d's proof differs from the one-rank proof" % r
            assert rc == 0
            assert n_vec == 1, "the index-cyclic inner-product argument was not taken (vector gathers: %d)" % n_vec
            # three commitment MSMs (single phase) + L and R of the nine partitioned rounds 2^22 -> 2^13 (the frozen tail is replicated)
            assert n_pts >= 3 + 2 * 9, "too few point reduces for a partitioned proof: %d" % n_pts
    finally:
        single.close()


def test_batch_of_600_distinct_statements(eng):
    """one template per STRUCTURE, not per statement: 600 distinct-witness proofs (square chains with 600 different public outputs,
    range proofs of 64 different values) in one batch — more than the template cache (64) and more than a 512-block"""
    from ark_bulletproofs_amd import engine as E

    inst = []
    for i in range(536):
        pr = eng.prove_scenario(E.SC_SQUARE_CHAIN, [8, 0], bytes([i & 255, i >> 8, 7] + [2] * 29), m_cap=8)
        inst.append((E.SC_SQUARE_CHAIN, [8, 0], pr.proof, pr.commitments, pr.publics))
    for i in range(64):
        pr = eng.prove_scenario(E.SC_RANGE, [16, 1000 + i], bytes([i, 9] + [2] * 30), m_cap=8)
        inst.append((E.SC_RANGE, [16, 1000 + i], pr.proof, pr.commitments, pr.publics))
    assert len({i[4].tobytes() for i in inst[:536]}) > 500       # the public outputs really differ
    rc, _ = eng.batch_verify(inst, bytes([6]) * 32)
    assert rc == 0
    sc, prm, proof, cm, pb = inst[530]
    inst[530] = (sc, prm, proof, cm, inst[529][4])              # another statement's public output
    rc, _ = eng.batch_verify(inst, bytes([6]) * 32)
    assert rc == -4


def test_cfg3_shuffle_statement_full_size():
    """BASELINE cfg3's first variant: k-shuffle with k = 2^19 + 1 -> 2^20 phase-2 multipliers, m = 2^20 + 2 commitments (made in one
    GPU batch, bp_stmt_prover_create_dev), q = 2^21 + 1.  prove -> verify -> tamper; the commitments of a sample of inputs are
    checked against the oracle-verified small-batch path."""
    import ark_bulletproofs_amd as A
    from ark_bulletproofs_amd import engine as E

    k = (1 << 19) + 1
    e = A.Engine(curve=0)
    try:
        e.gens_derive(1 << 20)
        seed = bytes([3]) * 32
        st = E.Statement(0, E.SC_SHUFFLE, [k], seed, engine=e)
        commits, pubs, _, _ = st.info(m_cap=2 * k + 8)
        assert len(commits) == 2 * k
        # the shuffle's outputs are a rotation of its inputs with fresh blindings: all 2k commitments distinct
        assert len({c.tobytes() for c in commits[:: max(1, k // 512)]}) == len(commits[:: max(1, k // 512)])
        proof, _ = st.prove(e)
        assert len(proof) == 539 + 66 * 20
        assert e.verify_scenario(E.SC_SHUFFLE, [k], proof, commits, pubs) == 0
        bad = bytearray(proof)
        bad[11 * 33 + 5] ^= 1   # t_x
        assert e.verify_scenario(E.SC_SHUFFLE, [k], bytes(bad), commits, pubs) == -4
        swapped = commits.copy()
        swapped[[0, 1]] = swapped[[1, 0]]   # a different statement: inputs permuted without the outputs following
        assert e.verify_scenario(E.SC_SHUFFLE, [k], proof, swapped, pubs) == -4
    finally:
        e.close()


def test_zorro_prove_verify_roundtrip_2pow16_and_batch():
    """the second curve of cfg5 (zorro: a = 6, 255-bit scalar field, no endomorphism -> plain NAF fold ladder) at full-size
    kernels: prove -> verify -> tamper at 2^16, and a mixed batch through the block pipeline"""
    import ark_bulletproofs_amd as A

    e = A.Engine(curve=1)
    try:
        N = 1 << 16
        e.gens_derive(N)
        pr = e.prove_scenario(3, [N, 0], bytes([6]) * 32, m_cap=8)
        assert len(pr.proof) == 539 + 66 * 16
        assert e.verify_scenario(3, [N, 0], pr.proof, pr.commitments, pr.publics) == 0
        bad = bytearray(pr.proof)
        bad[11 * 33 + 70] ^= 1   # e_blinding
        assert e.verify_scenario(3, [N, 0], bytes(bad), pr.commitments, pr.publics) == -4
        inst = [(3, [N, 0], pr.proof, pr.commitments, pr.publics)]
        for i in range(3):
            p2 = e.prove_scenario(4, [64, 32, 0], bytes([7, i] + [7] * 30), m_cap=72)
            inst.append((4, [64, 32, 0], p2.proof, p2.commitments, p2.publics))
        rc, _ = e.batch_verify(inst, bytes([8]) * 32)
        assert rc == 0
        inst[2] = (4, [64, 32, 0], inst[3][2], inst[2][3], inst[2][4])   # proof of another statement
        rc, _ = e.batch_verify(inst, bytes([8]) * 32)
        assert rc == -4
    finally:
        e.close()


def test_bench_configuration_2pow20_with_tables_checked():
    """The configuration bench.py times: N = 2^20 on secq256k1 with the fold tables for the first two rounds
    at the automatically chosen window width (3N/4 bases, w = 8: 219 GB) and the fixed-base MSM rows (8.7 GB).  (a) every entry of both kinds of
    table passes the chain-rule check, and a single corrupted entry — anywhere, here deep inside the tables — turns it red;
    (b) prove -> verify -> tamper with the tables in use; (c) the same statement proved with the tables released gives the same
    proof bytes."""
    import ark_bulletproofs_amd as A
    from ark_bulletproofs_amd import engine as E

    N, cv = 1 << 20, 0
    e = A.Engine(curve=cv)
    try:
        e.gens_derive(N)
        assert e.gens_msm_tables(N) > 0
        # bench.py's table settings: bases [0, 3N/4) (the first TWO fold rounds come from the tables), the widest window that fits
        # what is free after the MSM rows, one proof's workspaces and 12 GB of slack — w = 8, 219 GB on a 288 GB MI355X
        wbits, nbytes = e.gens_fold_tables(N * 3 // 4, window_bits=0, budget_bytes=288 * 10**9 - 9 * 10**9 - 3000 * N - (12 << 30) - (8 << 30))
        assert wbits == 8 and nbytes > 200e9
        assert e.gens_tables_check() == (0, 0)
        # corrupt ONE coordinate word of one entry far inside each table, check, restore, check
        for which, slot in ((0, 0), (1, 0), (2, 1), (3, 1)):
            dptr, size = e.debug_tables_ptr(which)
            assert dptr and size >= 64
            off = (size // 64 * 5 // 7) * 64 + 24
            orig = e.debug_poke(dptr + off, nbytes=4)
            flipped = orig.copy()
            flipped[1] ^= 0x10
            e.debug_poke(dptr + off, flipped)
            bad = e.gens_tables_check()
            assert bad[slot] >= 1 and bad[1 - slot] == 0, (which, bad)
            e.debug_poke(dptr + off, orig)
        assert e.gens_tables_check() == (0, 0)
        seed = bytes([3, 0, 0, 0, 0]) + bytes([3]) * 27                      # bench.py's statement_seed(0, 0)
        st = E.Statement(cv, E.SC_SQUARE_CHAIN, [N, 0], seed)
        commits, pubs, nm, nq = st.info(m_cap=8)
        proof, _ = st.prove(e)
        st.free()
        assert e.verify_scenario(E.SC_SQUARE_CHAIN, [N, 0], proof, commits, pubs) == 0
        bad = bytearray(proof)
        bad[11 * 33 + 9] ^= 1          # t_x
        assert e.verify_scenario(E.SC_SQUARE_CHAIN, [N, 0], bytes(bad), commits, pubs) == -4
        # a 2-phase statement of the same size through the same tables (all G factors = u)
        k = (1 << 19) + 1
        st2 = E.Statement(cv, E.SC_SHUFFLE, [k], seed, engine=e)
        commits2, pubs2, _, _ = st2.info(m_cap=2 * k + 8)
        proof2, _ = st2.prove(e)
        st2.free()
        assert e.verify_scenario(E.SC_SHUFFLE, [k], proof2, commits2, pubs2) == 0
        # tables off: byte-identical proofs
        e.gens_fold_tables(0)
        e.gens_msm_tables(0)
        st = E.Statement(cv, E.SC_SQUARE_CHAIN, [N, 0], seed)
        proof_off, _ = st.prove(e)
        st.free()
        assert proof_off == proof, "the 2^20 proof made with the tables differs from the one made without them"
        st2 = E.Statement(cv, E.SC_SHUFFLE, [k], seed, engine=e)
        proof2_off, _ = st2.prove(e)
        st2.free()
        assert proof2_off == proof2
    finally:
        e.close()


@pytest.mark.parametrize("curve", [0, 1], ids=["secq256k1", "zorro"])
def test_auto_tables_2pow14_against_oracle_bytes(oracle, curve):
    """the bench's table settings (automatic window width, fixed-base MSM rows in use) at a size the oracle proves in seconds:
    proof bytes equal the oracle's for a 1-phase and a 2-phase statement; the tables pass the chain-rule check"""
    import ark_bulletproofs_amd as A

    N = 1 << 14
    e = A.Engine(curve=curve)
    try:
        e.gens_derive(N)
        wbits, _ = e.gens_fold_tables(N // 2, window_bits=0)
        assert wbits == 8
        e.gens_msm_tables(N)
        e.set_tuning(5, 4096)          # BP_TUNE_MSM_FIXED_MIN: the fixed-base schedule at these sizes too
        assert e.gens_tables_check() == (0, 0)
        for sc, prm, mcap in [(3, [N, 0], 8), (0, [N // 2 + 1], N + 16)]:
            ref = oracle.r1cs_prove(curve, sc, prm, bytes([7]) * 32, N, m_cap=mcap)
            assert ref.rc == 0
            got = e.prove_scenario(sc, prm, bytes([7]) * 32, m_cap=mcap)
            assert got.proof == ref.proof, (curve, sc)
            assert e.verify_scenario(sc, prm, got.proof, got.commitments, got.publics) == 0
    finally:
        e.close()
