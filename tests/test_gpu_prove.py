"""GPU parity of r1cs::Prover::prove (src/r1cs/prover.rs:454-831): for the reference's own test/bench
statements the MI355X engine must emit byte-identical proofs (and commitments) to the CPU oracle under
the same external ChaCha20 seed, and the oracle's verifier must accept them."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = bytes([7]) * 32


@pytest.fixture(scope="module", params=[0, 1], ids=["secq256k1", "zorro"])
def eng(request):
    import ark_bulletproofs_amd as A

    e = A.Engine(curve=request.param)
    e.gens_derive(128)
    yield e
    e.close()


def test_derived_generators_resident(eng, oracle):
    G, H = eng.gens_download(128)
    Go, Ho = oracle.bp_gens(eng.curve, 128)
    assert (G == Go).all() and (H == Ho).all()


CASES = [
    (0, [1]), (0, [2]), (0, [3]), (0, [4]), (0, [5]), (0, [6]), (0, [7]), (0, [24]), (0, [42]),
    (2, [3, 4, 6, 1, 40, 9]), (2, [3, 4, 6, 1, 40, 10]),
    (1, [2, 3]), (1, [10, 1000]), (1, [32, 77]), (1, [63, (1 << 63) - 5]), (1, [10, 1024]),
    (3, [13, 0]), (3, [100, 0]), (3, [13, 1]), (4, [3, 8, 0]), (4, [3, 8, 1]),
]


@pytest.mark.parametrize("sc,prm", CASES)
def test_prove_matches_oracle_bytes(eng, oracle, sc, prm):
    O, cv = oracle, eng.curve
    ref = O.r1cs_prove(cv, sc, prm, SEED, 128, m_cap=128)
    assert ref.rc == 0
    got = eng.prove_scenario(sc, prm, SEED, m_cap=128)
    assert (got.commitments == ref.commitments).all()
    assert (got.publics == ref.publics).all()
    assert got.proof == ref.proof
    # the reference verifier logic (oracle) decides accept / reject exactly as for its own proof
    rc_ref = O.r1cs_verify(cv, sc, prm, 128, ref.proof, ref.commitments, ref.publics)
    assert O.r1cs_verify(cv, sc, prm, 128, got.proof, got.commitments, got.publics) == rc_ref


@pytest.mark.parametrize("freeze", [0, 8])
@pytest.mark.parametrize("sc,prm", [(0, [24]), (1, [32, 77]), (3, [100, 0]), (4, [3, 8, 0])])
def test_prove_large_round_kernels_at_small_sizes(oracle, sc, prm, freeze):
    """prover with the large-input kernels forced (shared-inversion fold epilogue, two-level MSM sort), generator folds to the
    end (freeze = 0) or coefficient folds over frozen vectors from length 8 on: same proof bytes"""
    import ark_bulletproofs_amd as A

    for cv in (0, 1):
        e = A.Engine(curve=cv)
        e.gens_derive(128)
        e.set_tuning(12, 0)           # BP_TUNE_DIRECT_MAX: not the small-statement path (tests/test_gpu_small.py) — the schedules below
        e.set_tuning(0, 1)
        e.set_tuning(1, 1)
        e.set_tuning(2, freeze)
        e.set_tuning(3, 1 if freeze else 1 << 18)   # running-sum window aggregation at small sizes too
        ref = oracle.r1cs_prove(cv, sc, prm, SEED, 128, m_cap=128)
        got = e.prove_scenario(sc, prm, SEED, m_cap=128)
        assert got.proof == ref.proof and (got.commitments == ref.commitments).all()
        e.close()


def test_prove_needs_enough_generators(eng):
    import ark_bulletproofs_amd as A

    with pytest.raises(A.ArkbpError) as ei:
        eng.prove_scenario(3, [200, 0], SEED)  # 200 multipliers -> padded 256 > capacity 128
    assert ei.value.code == -5  # InvalidGeneratorsLength


def test_statement_api_and_concurrent_proofs(eng, oracle):
    """bp_stmt_*: setup separated from prove(); several proofs in flight on one GPU (own ctx + host thread each, shared
    resident generator tables) must still be byte-identical to the oracle's proofs."""
    import threading

    import ark_bulletproofs_amd as A
    from ark_bulletproofs_amd import engine as E

    O, cv = oracle, eng.curve
    cases = [(3, [100, 0]), (0, [24]), (1, [32, 77]), (4, [3, 8, 0]), (3, [64, 0]), (0, [7])]
    engs = [A.Engine(curve=cv) for _ in cases]
    for e in engs:
        e.share_gens_from(eng)
    stmts = [E.Statement(cv, sc, prm, bytes([50 + i]) * 32) for i, (sc, prm) in enumerate(cases)]
    cm0, pb0, nm, nq = stmts[0].info()
    assert nm == 100 and nq == 201 and len(cm0) == 1
    out = [None] * len(cases)

    def work(k):
        out[k] = stmts[k].prove(engs[k])

    th = [threading.Thread(target=work, args=(k,)) for k in range(len(cases))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i, (sc, prm) in enumerate(cases):
        ref = O.r1cs_prove(cv, sc, prm, bytes([50 + i]) * 32, 128, m_cap=128)
        assert out[i] is not None and out[i][0] == ref.proof
    with pytest.raises(A.ArkbpError):  # Prover::prove consumes self
        stmts[0].prove(engs[0])
    for e in engs:
        e.close()


def test_batched_precompute_x8(eng, oracle):
    """bp_stmt_precompute_batch: 8 same-shaped statements share one AVX-512 Keccak-f x8 stream for their TranscriptRng
    chains; the proofs must still be byte-identical to the oracle's (which runs plain scalar merlin)."""
    from ark_bulletproofs_amd import engine as E

    O, cv = oracle, eng.curve
    sc, prm = 3, [100, 0]
    stmts = [E.Statement(cv, sc, prm, bytes([70 + i]) * 32) for i in range(8)]
    odd = E.Statement(cv, 1, [16, 5], bytes([90]) * 32)          # different shape: takes the scalar path
    E.precompute_batch(stmts + [odd])
    for i, st in enumerate(stmts):
        proof, _ = st.prove(eng)
        assert proof == O.r1cs_prove(cv, sc, prm, bytes([70 + i]) * 32, 128, m_cap=8).proof
    proof, _ = odd.prove(eng)
    assert proof == O.r1cs_prove(cv, 1, [16, 5], bytes([90]) * 32, 128).proof


@pytest.mark.parametrize("world", [2, 3])
def test_window_sharded_prover_ranks_as_threads(oracle, world):
    """north_star / cfg5 partition: every rank runs the same prove() and verify(), each MSM inside accumulates only the rank's
    Pippenger windows, partial points are summed through the host collective (here: an in-process all-gather between threads,
    one Engine per rank on the same GPU).  Every rank must emit the oracle's proof bytes and accept it."""
    import threading

    import ark_bulletproofs_amd as A
    from ark_bulletproofs_amd import engine as E
    from ark_bulletproofs_amd import parallel as P

    for cv in (0, 1):
        cases = [(0, [7]), (3, [100, 0]), (4, [3, 8, 0])]
        refs = [oracle.r1cs_prove(cv, sc, prm, SEED, 128, m_cap=128) for sc, prm in cases]
        bar = threading.Barrier(world)
        slots = [None] * world
        out = [None] * world
        errors = []

        def run(rank):
            try:
                e = A.Engine(curve=cv)
                e.gens_derive(128)

                def allgather(arr):
                    slots[rank] = np.array(arr, copy=True)
                    bar.wait()
                    res = np.stack(slots)
                    bar.wait()
                    return res

                P.enable_window_sharding(e, cv, E.host_points_sum, rank, world, allgather=allgather)
                # world 2: the prover also partitions the inner-product argument index-cyclically (thresholds lowered so that the
                # 128- and 32-element IPAs of these statements take that path: cyclic rounds down to 4 elements, gather, frozen tail);
                # world 3 is not a power of two: folds stay replicated
                e.set_tuning(4, 16)   # BP_TUNE_CYCLIC_MIN
                e.set_tuning(2, 4)    # BP_TUNE_IPA_FREEZE_LEN
                got = []
                for (sc, prm), ref in zip(cases, refs):
                    pr = e.prove_scenario(sc, prm, SEED, m_cap=128)
                    rc = e.verify_scenario(sc, prm, pr.proof, pr.commitments, pr.publics)
                    bad = bytearray(pr.proof)
                    bad[-1] ^= 1
                    rc_bad = e.verify_scenario(sc, prm, bytes(bad), pr.commitments, pr.publics)
                    got.append((pr.proof, rc, rc_bad))
                out[rank] = got
                P.enable_window_sharding(e, cv, E.host_points_sum, 0, 1)
                e.close()
            except Exception as ex:
                errors.append(ex)
                bar.abort()

        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        for r in range(world):
            for (proof, rc, rc_bad), ref in zip(out[r], refs):
                assert proof == ref.proof and rc == 0 and rc_bad in (-4, -6)


@pytest.mark.parametrize("tables", ["one_round", "two_rounds", "two_rounds_sliced"])
def test_window_and_cyclic_sharded_prover_2pow16_ranks_as_threads(tables):
    two_rounds = tables != "one_round"
    """cfg5's partition at full-size kernels with the default thresholds: a 2^16-constraint proof by two ranks (threads, one Engine
    each on the same GPU): Pippenger windows of the commitment MSMs partitioned, the IPA index-cyclic (32768 elements per rank,
    gather at 1024), the verifier's mega-check window-sharded.  Both ranks must emit exactly the single-GPU proof."""
    import threading

    import ark_bulletproofs_amd as A
    from ark_bulletproofs_amd import engine as E
    from ark_bulletproofs_amd import parallel as P

    N, world, cv = 1 << 16, 2, 0
    cases = [(3, [N, 0]), (0, [(N // 2) + 1])]      # square chain (geometric H factors, constant G factors) and a 2^16-multiplier shuffle (all phase 2)
    single = A.Engine(curve=cv)
    single.gens_derive(N)
    refs = [single.prove_scenario(sc, prm, SEED, m_cap=N + 16) for sc, prm in cases]
    # the ranks share the first-round fold tables and the fixed-base MSM rows too: the cyclic slices index both with a stride, the
    # commitments take the fixed-base schedule over per-rank blocks of the terms (round 4: the sharded prover keeps the single-GPU
    # MSM algorithm; BP_TUNE_MSM_FIXED_MIN lowered on every rank so that 2^16 reaches it): same proofs
    # two_rounds: fold tables over 3N/4 bases — every rank DEFERS the first fold of its slice and takes its second fold straight from
    # the tables, the round in between runs its L / R over the slice with split scalars (the single-GPU schedule of round 3, on slices)
    _, whole_bytes = single.gens_fold_tables(N * 3 // 4 if two_rounds else N // 2, window_bits=4)
    single.gens_msm_tables(N)
    single.set_tuning(5, 4096)
    for (sc, prm), ref in zip(cases, refs):
        assert single.verify_scenario(sc, prm, ref.proof, ref.commitments, ref.publics) == 0
    bar = threading.Barrier(world)
    slots, out, errors = [None] * world, [None] * world, []
    big_gathers = [0] * world

    def run(rank):
        try:
            e = A.Engine(curve=cv)
            e.share_gens_from(single)
            if tables == "two_rounds_sliced":
                # 1/world of the fold tables per rank: only the generators rank + i * world — all its slice of the argument looks up
                wb, nbytes = e.gens_fold_tables(N * 3 // 4, window_bits=4, rank=rank, world=world)
                assert wb == 4 and nbytes * world == whole_bytes
                assert e.gens_tables_check()[0] == 0

            def allgather(arr):
                if np.asarray(arr).size > 8:
                    big_gathers[rank] += 1       # not a 64-byte point: the IPA's vector gather
                slots[rank] = np.array(arr, copy=True)
                bar.wait()
                res = np.stack(slots)
                bar.wait()
                return res

            P.enable_window_sharding(e, cv, E.host_points_sum, rank, world, allgather=allgather)
            e.set_tuning(5, 4096)       # BP_TUNE_MSM_FIXED_MIN
            got = []
            for (sc, prm) in cases:
                pr = e.prove_scenario(sc, prm, SEED, m_cap=N + 16)
                got.append((pr.proof, e.verify_scenario(sc, prm, pr.proof, pr.commitments, pr.publics)))
            out[rank] = got
            fb_sharded[rank] = e.msm_stats()[1]
            fold_stats[rank] = e.fold_stats()
            P.enable_window_sharding(e, cv, E.host_points_sum, 0, 1)
            e.close()
        except Exception as ex:
            errors.append(ex)
            bar.abort()

    fb_sharded = [0] * world
    fold_stats = [None] * world
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for r in range(world):
        for (proof, rc), ref in zip(out[r], refs):
            assert proof == ref.proof and rc == 0
        assert big_gathers[r] == len(cases), "the index-cyclic IPA path was not taken"
        # (at this size the fixed-base schedule declines on every rank — its bins are laid out for >= 2^20 terms — and each rank falls
        # back to the ordinary MSM over ITS block / slice: the ranks still meet in one reduce per MSM, which is what this run checks;
        # the schedule itself is asserted at 2^22 in tests/test_gpu_fullsize.py)
        assert fb_sharded[r] >= 0
        # one deferred first fold and one second fold from the tables per proof and rank — or none with tables for one round only
        assert fold_stats[r] == ((len(cases), len(cases)) if two_rounds else (0, 0)), fold_stats[r]
    single.close()


@pytest.mark.parametrize("w", [0, 3, 8])
def test_prove_with_first_round_fold_tables(oracle, w):
    """bp_gens_fold_tables: the first fold round through fixed-base tables of the generators (window widths 3, 8 and the
    automatic choice) must leave the proofs byte-identical to the oracle's — square chains (constant G factors, geometric H
    factors), shuffles (all multipliers randomized: G factors all u) and multi-range circuits, 128 to 1024 multipliers, with the
    shared-inversion epilogue both off and forced on; a statement whose left half is larger than the tables falls back to the ladder"""
    import ark_bulletproofs_amd as A

    for cv in (0, 1):
        e = A.Engine(curve=cv)
        e.gens_derive(1024)
        e.set_tuning(12, 0)           # BP_TUNE_DIRECT_MAX: not the small-statement path (tests/test_gpu_small.py) — the schedules below
        wb, nbytes = e.gens_fold_tables(256, window_bits=w)
        assert 2 <= wb <= 8 and nbytes > 0 and (w == 0 or wb == w)
        for batch_min in (65536, 1):
            e.set_tuning(0, batch_min)      # BP_TUNE_FOLD_BATCH_MIN: in-lane inversions / the shared-inversion epilogue
            for sc, prm in [(3, [100, 0]), (3, [500, 0]), (0, [65]), (0, [200]), (4, [16, 16, 0]), (3, [1000, 0])]:
                ref = oracle.r1cs_prove(cv, sc, prm, SEED, 1024, m_cap=512)
                got = e.prove_scenario(sc, prm, SEED, m_cap=512)
                assert got.proof == ref.proof, (cv, w, sc, prm)
        # the tables go along with bp_gens_share
        e2 = A.Engine(curve=cv)
        e2.set_tuning(12, 0)
        e2.share_gens_from(e)
        ref = oracle.r1cs_prove(cv, 3, [300, 0], SEED, 1024, m_cap=8)
        assert e2.prove_scenario(3, [300, 0], SEED, m_cap=8).proof == ref.proof
        e2.close()
        e.gens_fold_tables(0)
        assert e.prove_scenario(3, [300, 0], SEED, m_cap=8).proof == ref.proof
        e.close()


@pytest.mark.parametrize("fold_tables", [False, True])
def test_prove_with_fixed_base_msm_tables(oracle, fold_tables):
    """bp_gens_msm_tables: the commitment MSMs and the first round's L / R as fixed-base MSMs over precomputed rows of the
    generator tables (one bucket set for all Pippenger windows) — proofs byte-identical to the oracle's, alone and together with
    the first-round fold tables; 0/1 witness vectors (range proofs) overflow a bin and take the ordinary schedule"""
    import ark_bulletproofs_amd as A

    for cv in (0, 1):
        e = A.Engine(curve=cv)
        e.gens_derive(4096)
        e.set_tuning(12, 0)           # BP_TUNE_DIRECT_MAX: not the small-statement path (tests/test_gpu_small.py) — the schedules below
        assert e.gens_msm_tables(4096) > 0
        e.set_tuning(5, 4096)      # BP_TUNE_MSM_FIXED_MIN: take the fixed-base schedule at these sizes
        if fold_tables:
            e.gens_fold_tables(2048, window_bits=4)
        for sc, prm, mcap in [(3, [3000, 0], 8), (3, [4096, 0], 8), (0, [2049], 4200), (4, [64, 64, 0], 72)]:
            ref = oracle.r1cs_prove(cv, sc, prm, SEED, 4096, m_cap=mcap)
            got = e.prove_scenario(sc, prm, SEED, m_cap=mcap)
            assert got.proof == ref.proof, (cv, sc, prm)
        e.gens_msm_tables(0)
        e.close()


@pytest.mark.parametrize("w", [0, 4])
def test_prove_two_fold_rounds_from_the_tables(oracle, w):
    """Fold tables that cover the bases [0, 3N/4): the prover defers its first fold and produces the second round's vectors
    straight from the tables (k_ipa_fold_tab2), with that round's L / R as MSMs over the generator tables with split scalars
    (k_ipa_scalars_deferred; fixed-base rows or the ordinary schedule).  Proof bytes must equal the oracle's for single-phase
    (square chain: constant G factors, geometric H factors), two-phase (shuffle: all factors u) and multi-range circuits, with
    and without the fixed-base MSM rows, the shared-inversion epilogue on and off; a statement larger than the tables' reach
    falls back to the single-round table fold."""
    import ark_bulletproofs_amd as A

    for cv in (0, 1):
        e = A.Engine(curve=cv)
        try:
            N = 4096
            e.gens_derive(N)
            e.set_tuning(12, 0)           # BP_TUNE_DIRECT_MAX: not the small-statement path (tests/test_gpu_small.py) — the schedules below
            wb, nbytes = e.gens_fold_tables(N * 3 // 4, window_bits=w)
            assert nbytes > 0 and (w == 0 or wb == w)
            e.set_tuning(2, 16)        # BP_TUNE_IPA_FREEZE_LEN: the frozen tail starts at 16 so that small statements defer their first fold
            assert e.gens_tables_check()[0] == 0
            cases = [(3, [4096, 0], 8), (3, [2048, 0], 8), (0, [2049], 4200), (0, [513], 1100), (4, [64, 64, 0], 72), (3, [1024, 0], 8), (3, [1000, 0], 8)]
            refs = {}
            for sc, prm, mcap in cases:
                refs[(sc, tuple(prm))] = oracle.r1cs_prove(cv, sc, prm, SEED, N, m_cap=mcap)
            for fixed_rows in (False, True):
                if fixed_rows:
                    e.gens_msm_tables(N)
                    e.set_tuning(5, 4096)
                for batch_min in (65536, 1):
                    e.set_tuning(0, batch_min)
                    for sc, prm, mcap in cases:
                        got = e.prove_scenario(sc, prm, SEED, m_cap=mcap)
                        assert got.proof == refs[(sc, tuple(prm))].proof, (cv, w, fixed_rows, batch_min, sc, prm)
                        assert e.verify_scenario(sc, prm, got.proof, got.commitments, got.publics) == 0
            # tables for only N/2 bases again: the single-round table fold
            e.gens_fold_tables(N // 2, window_bits=4)
            sc, prm, mcap = cases[0]
            assert e.prove_scenario(sc, prm, SEED, m_cap=mcap).proof == refs[(sc, tuple(prm))].proof
        finally:
            e.close()


def test_prove_with_quad_cooperative_fold_rounds(oracle):
    """BP_TUNE_FOLD_QUAD_MAX: the fold rounds below the threshold run with four lanes per point (ecq.cuh: k_ipa_fold_glv<.., true> on
    secq256k1, k_ipa_fold_uniform<.., true> on zorro); proofs stay byte-identical to the oracle's, with the frozen tail moved down so
    that several rounds take the quad form, and with the shared-inversion epilogue forced on and off"""
    import ark_bulletproofs_amd as A

    for cv in (0, 1):
        e = A.Engine(curve=cv)
        try:
            e.gens_derive(2048)
            e.set_tuning(12, 0)           # BP_TUNE_DIRECT_MAX: not the small-statement path (tests/test_gpu_small.py) — the schedules below
            e.set_tuning(8, 1 << 12)     # BP_TUNE_FOLD_QUAD_MAX
            e.set_tuning(2, 8)           # BP_TUNE_IPA_FREEZE_LEN
            for batch_min in (65536, 1):
                e.set_tuning(0, batch_min)
                for sc, prm, mcap in [(3, [2048, 0], 8), (0, [600], 1300), (1, [64, 12345], 8), (3, [300, 0], 8)]:
                    ref = oracle.r1cs_prove(cv, sc, prm, SEED, 2048, m_cap=mcap)
                    got = e.prove_scenario(sc, prm, SEED, m_cap=mcap)
                    assert got.proof == ref.proof, (cv, batch_min, sc, prm)
        finally:
            e.close()
