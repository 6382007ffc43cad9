"""The small-statement path of the prover (include/arkbp.h BP_TUNE_DIRECT_MAX, csrc/small.cuh): for padded sizes up to BP_TUNE_DIRECT_MAX (default 8192) every MSM of
Prover::prove (src/r1cs/prover.rs:516-649) and every round of InnerProductProof::create (src/inner_product_proof.rs:86-213) is a sum
over direct window tables of the first generators, and G / H are never folded.  The proofs must be byte-identical to the oracle's and
to the folding schedule's — the reference's own benchmark range (benches/r1cs_secq256k1.rs:152-250: k-shuffles, k = 2 .. 1024)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = bytes([11]) * 32
DIRECT_MAX = 12   # BP_TUNE_DIRECT_MAX


@pytest.fixture(scope="module", params=[0, 1], ids=["secq256k1", "zorro"])
def eng(request):
    import ark_bulletproofs_amd as A

    e = A.Engine(curve=request.param)
    e.gens_derive(4096)
    yield e
    e.close()


# the reference bench's shuffles (two-phase: every multiplier is allocated in the randomized phase), range proofs (0/1 witness
# columns: most table look-ups are the digit 1 or skipped), the example gadget, square chains (single-phase) and multi-range circuits
CASES = [
    (0, [2], 16), (0, [3], 16), (0, [4], 16), (0, [8], 32), (0, [33], 80), (0, [64], 136), (0, [1024], 2056),
    (1, [8, 200], 8), (1, [64, (1 << 64) - 1], 8), (1, [64, 0], 8),
    (2, [3, 4, 6, 1, 40, 9], 8),
    (3, [1, 0], 8), (3, [2, 0], 8), (3, [100, 0], 8), (3, [2048, 0], 8), (3, [4096, 0], 8), (3, [13, 1], 8),
    (4, [3, 8, 0], 16), (4, [32, 64, 0], 40), (4, [3, 8, 1], 16),
]


@pytest.mark.parametrize("sc,prm,mcap", CASES)
def test_direct_tables_prove_matches_oracle_and_folding_schedule(eng, oracle, sc, prm, mcap):
    cv = eng.curve
    ref = oracle.r1cs_prove(cv, sc, prm, SEED, 4096, m_cap=mcap)
    assert ref.rc == 0
    eng.set_tuning(DIRECT_MAX, 4096)
    runs0, _ = eng.direct_stats()
    got = eng.prove_scenario(sc, prm, SEED, m_cap=mcap)
    runs1, cap = eng.direct_stats()
    assert cap == 4096 and runs1 > runs0          # the path under test really ran
    assert got.proof == ref.proof and (got.commitments == ref.commitments).all() and (got.publics == ref.publics).all()
    # (the last parameter of scenarios 3 and 4 plants a wrong witness: those proofs must FAIL, identically on both sides)
    rc_ref = oracle.r1cs_verify(cv, sc, prm, 4096, ref.proof, ref.commitments, ref.publics)
    assert rc_ref == (0 if not (sc in (3, 4) and prm[-1]) else oracle.E_VERIFICATION)
    assert (eng.verify_scenario(sc, prm, got.proof, got.commitments, got.publics) == 0) == (rc_ref == 0)
    # the folding schedule (ladders / frozen tail / bucket MSMs) on the same ctx
    eng.set_tuning(DIRECT_MAX, 0)
    try:
        runs2, _ = eng.direct_stats()
        old = eng.prove_scenario(sc, prm, SEED, m_cap=mcap)
        assert eng.direct_stats()[0] == runs2
        assert old.proof == got.proof
    finally:
        eng.set_tuning(DIRECT_MAX, 4096)


def test_direct_tables_limit_and_rebuild(oracle):
    """the tables cover min(generators, BP_TUNE_DIRECT_MAX) bases per vector and are built by the first proof that needs them; a
    statement beyond them takes the folding schedule; installing other generators drops them; a knob above 2^16 is refused"""
    import ark_bulletproofs_amd as A

    e = A.Engine(curve=0)
    try:
        e.gens_derive(512)
        e.set_tuning(DIRECT_MAX, 128)
        assert e.direct_stats() == (0, 0)
        ref = oracle.r1cs_prove(0, 3, [100, 0], SEED, 512, m_cap=8)
        assert e.prove_scenario(3, [100, 0], SEED, m_cap=8).proof == ref.proof
        runs, cap = e.direct_stats()
        assert cap == 128 and runs == 3 + 5 + 2 * 7      # A_I, A_O, S in one launch, T_1, T_3 .. T_6 in one, then L and R of 7 rounds
        big = oracle.r1cs_prove(0, 3, [200, 0], SEED, 512, m_cap=8)     # padded 256 > 128
        assert e.prove_scenario(3, [200, 0], SEED, m_cap=8).proof == big.proof
        assert e.direct_stats()[0] == runs
        e.set_tuning(DIRECT_MAX, 512)                                   # a longer reach: rebuilt by the next proof
        assert e.prove_scenario(3, [200, 0], SEED, m_cap=8).proof == big.proof
        runs2, cap2 = e.direct_stats()
        assert cap2 == 512 and runs2 > runs
        e.gens_derive(256)                                              # other tables (here: shorter): the direct tables go
        assert e.direct_stats()[1] == 0
        assert e.prove_scenario(3, [200, 0], SEED, m_cap=8).proof == big.proof
        assert e.direct_stats()[1] == 256
        with pytest.raises(A.ArkbpError):
            e.set_tuning(DIRECT_MAX, (1 << 16) + 1)
        # the explicit form: built ahead of the first proof (e.g. before bp_gens_share), freed with 0
        assert e.gens_direct_tables(0) == 0 and e.direct_stats()[1] == 0
        assert e.gens_direct_tables(64) == (2 + 2 * 64) * 64 * 15 * 64 and e.direct_stats()[1] == 64
        runs3 = e.direct_stats()[0]
        small = oracle.r1cs_prove(0, 3, [50, 0], SEED, 512, m_cap=8)
        assert e.prove_scenario(3, [50, 0], SEED, m_cap=8).proof == small.proof       # padded 64: inside the tables as built
        assert e.direct_stats() == (runs3 + 3 + 5 + 2 * 6, 64)                        # A_I, A_O, S + T_1, T_3..T_6 + L, R of 6 rounds
        assert e.prove_scenario(3, [200, 0], SEED, m_cap=8).proof == big.proof        # padded 256: the next proof that needs more rebuilds them
        assert e.direct_stats()[1] == 256
    finally:
        e.close()


def test_direct_tables_go_along_with_shared_generators(eng, oracle):
    """bp_gens_share: a second ctx reads the first one's direct tables (several small proofs in flight on one GPU, one table set)"""
    import threading
    import ark_bulletproofs_amd as A

    cv = eng.curve
    eng.set_tuning(DIRECT_MAX, 4096)
    eng.prove_scenario(3, [10, 0], SEED, m_cap=8)            # (built)
    others = [A.Engine(curve=cv) for _ in range(3)]
    try:
        for o in others:
            o.share_gens_from(eng)
            assert o.direct_stats() == (0, 4096)
        cases = [(0, [16], 40), (3, [1000, 0], 8), (1, [32, 12345], 8)]
        refs = [oracle.r1cs_prove(cv, sc, prm, SEED, 4096, m_cap=mc) for sc, prm, mc in cases]
        out = [None] * 3

        def work(i):
            sc, prm, mc = cases[i]
            for _ in range(4):
                out[i] = others[i].prove_scenario(sc, prm, SEED, m_cap=mc)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for i in range(3):
            assert out[i].proof == refs[i].proof
            assert others[i].direct_stats()[0] > 0
    finally:
        for o in others:
            o.close()


@pytest.mark.parametrize("two_phase", [False, True])
def test_direct_tables_random_gadgets_through_the_recorder(eng, oracle, two_phase):
    """bp_cs recorder (random sparse one- and two-phase gadgets, tests/gadgets.py): Prover::prove over the direct tables == the
    folding schedule == the oracle's Prover on the same gadget"""
    from ark_bulletproofs_amd import engine as E
    import test_gpu_cs_api as T

    F = T.GD.Field(oracle, eng.curve)
    for struct_seed in (21, 22):
        eng.set_tuning(DIRECT_MAX, 4096)
        runs0 = eng.direct_stats()[0]
        a, V, pubs = T.product_prove(E, eng, F, struct_seed, 9, 3, two_phase)
        assert eng.direct_stats()[0] > runs0
        eng.set_tuning(DIRECT_MAX, 0)
        try:
            b, V2, pubs2 = T.product_prove(E, eng, F, struct_seed, 9, 3, two_phase)
        finally:
            eng.set_tuning(DIRECT_MAX, 4096)
        ref, Vo, pubs_o = T.oracle_prove(oracle, eng.curve, F, struct_seed, 9, 3, two_phase)
        assert a == b == ref and (V == V2).all() and (V == Vo).all() and pubs == pubs2 == pubs_o
