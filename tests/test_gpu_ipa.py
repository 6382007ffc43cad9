"""GPU parity of InnerProductProof::create (src/inner_product_proof.rs:37-239) through the C ABI: same
inputs and the same Merlin transcript (driven from the challenge callback) must give bit-identical
L_vec, R_vec, a, b as the CPU oracle; the oracle's verifier must accept the GPU proof."""
import numpy as np
import pytest

from test_oracle_protocol import _ipa_instance

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[0, 1], ids=["secq256k1", "zorro"])
def eng(request):
    import ark_bulletproofs_amd as A

    e = A.Engine(curve=request.param)
    yield e
    e.close()


def _challenger(O, cv, tr):
    def f(L, R):
        tr.append_point(cv, b"L", L)
        tr.append_point(cv, b"R", R)
        return tr.challenge_scalar(cv, b"u")

    return f


@pytest.mark.parametrize("n", [1, 2, 4, 32, 64, 256])
def test_ipa_create_matches_oracle(eng, oracle, n):
    O, cv = oracle, eng.curve
    G, H, Q, a, b, Gf, Hf, P = _ipa_instance(O, cv, n)
    # make G_factors non-trivial too (the R1CS prover passes [1]*n1 ++ [u]*(n2+pad))
    u = O.fe_rand(O.fid(cv, True), bytes([9]) * 32, 1)[0]
    Gf = Gf.copy()
    Gf[n // 2:] = u
    tr_o = O.Transcript(b"innerproducttest")
    tr_o.append_u64(b"n-marker", n)  # stand-in for innerproduct_domain_sep, identical on both sides
    Lo, Ro, ao, bo = None, None, None, None
    tr_ref = tr_o.clone()
    # oracle side: create() itself appends the domain separator, so feed the GPU callback a transcript
    # that has seen the same prefix
    Lo, Ro, ao, bo = O.ipa_create(cv, tr_ref, Q, Gf, Hf, G, H, a, b)
    tr_g = tr_o.clone()
    tr_g.append_message(b"dom-sep", b"ipp v1")
    tr_g.append_u64(b"n", n)
    Lg, Rg, ag, bg = eng.ipa_create(Q, Gf, Hf, G, H, a, b, _challenger(O, cv, tr_g))
    assert (Lg == Lo).all() and (Rg == Ro).all()
    assert (ag == ao).all() and (bg == bo).all()
    # both transcripts end in the same state
    assert tr_g.challenge_bytes(b"chk", 16) == tr_ref.challenge_bytes(b"chk", 16)


def test_ipa_proof_verifies(eng, oracle):
    O, cv = oracle, eng.curve
    n = 64
    G, H, Q, a, b, Gf, Hf, P = _ipa_instance(O, cv, n)
    tr = O.Transcript(b"innerproducttest")
    tr.append_message(b"dom-sep", b"ipp v1")
    tr.append_u64(b"n", n)
    L, R, ao, bo = eng.ipa_create(Q, Gf, Hf, G, H, a, b, _challenger(O, cv, tr))
    assert O.ipa_verify(cv, O.Transcript(b"innerproducttest"), n, Gf, Hf, P, Q, G, H, L, R, ao, bo) == 0


def test_ipa_rejects_bad_lengths(eng, oracle):
    O, cv = oracle, eng.curve
    G, H, Q, a, b, Gf, Hf, P = _ipa_instance(O, cv, 4)
    import ark_bulletproofs_amd as A

    with pytest.raises(ValueError):
        eng.ipa_create(Q, Gf, Hf, G, H, a[:3], b, lambda L, R: a[0])
    with pytest.raises(A.ArkbpError):  # n = 3 is not a power of two
        eng.ipa_create(Q, Gf[:3], Hf[:3], G[:3], H[:3], a[:3], b[:3], lambda L, R: a[0])


@pytest.mark.parametrize("n", [1, 2, 4, 32, 64])
def test_ipa_verify_on_gpu(eng, oracle, n):
    """InnerProductProof::verify (src/inner_product_proof.rs:321-382) on the GPU: same accept/reject as the oracle,
    on the reference's own make_ipp sizes (:530-553)."""
    O, cv = oracle, eng.curve
    FR = O.fid(cv, True)
    G, H, Q, a, b, Gf, Hf, P = _ipa_instance(O, cv, n)
    L, R, ao, bo = O.ipa_create(cv, O.Transcript(b"innerproducttest"), Q, Gf, Hf, G, H, a, b)
    # the caller's transcript replay yields the challenges
    tr = O.Transcript(b"innerproducttest")
    tr.append_message(b"dom-sep", b"ipp v1")
    tr.append_u64(b"n", n)
    ch = []
    for j in range(len(L)):
        tr.append_point(cv, b"L", L[j])
        tr.append_point(cv, b"R", R[j])
        ch.append(tr.challenge_scalar(cv, b"u"))
    ch = np.array(ch).reshape(-1, 4)
    assert eng.ipa_verify(n, Gf, Hf, P, Q, G, H, L, R, ch, ao, bo) == 0
    assert O.ipa_verify(cv, O.Transcript(b"innerproducttest"), n, Gf, Hf, P, Q, G, H, L, R, ao, bo) == 0
    bad_a = O.fe_op("add", FR, ao, O.fe_from_int(FR, 1))
    assert eng.ipa_verify(n, Gf, Hf, P, Q, G, H, L, R, ch, bad_a, bo) == -4
    assert eng.ipa_verify(n, Gf, Hf, G[0], Q, G, H, L, R, ch, ao, bo) == -4            # wrong P
    if n > 1:
        L2 = L.copy()
        L2[0] = R[0]
        assert eng.ipa_verify(n, Gf, Hf, P, Q, G, H, L2, R, ch, ao, bo) == -4
        assert eng.ipa_verify(n, Gf, Hf, P, Q, G, H, L[:-1], R[:-1], ch[:-1], ao, bo) == -4  # n != 2^lg_n
    # a proof created on the GPU verifies on the GPU
    tr_g = O.Transcript(b"innerproducttest")
    tr_g.append_message(b"dom-sep", b"ipp v1")
    tr_g.append_u64(b"n", n)
    chg = []

    def chal(Lp, Rp):
        tr_g.append_point(cv, b"L", Lp)
        tr_g.append_point(cv, b"R", Rp)
        u = tr_g.challenge_scalar(cv, b"u")
        chg.append(u)
        return u

    Lg, Rg, ag, bg = eng.ipa_create(Q, Gf, Hf, G, H, a, b, chal)
    assert eng.ipa_verify(n, Gf, Hf, P, Q, G, H, Lg, Rg, np.array(chg).reshape(-1, 4), ag, bg) == 0


@pytest.mark.parametrize("freeze", [0, 4, 16])
@pytest.mark.parametrize("n", [4, 64, 256])
def test_ipa_create_large_round_kernels_at_small_sizes(oracle, n, freeze):
    """the kernels large rounds switch to (shared-inversion fold epilogue k_ipa_fold_finish, two-level MSM sort) driven with
    small inputs through bp_ctx_set_tuning, with the generator folds kept to the end (freeze = 0) or replaced by coefficient
    folds over frozen vectors from length 4 / 16 on: same bytes as the oracle"""
    import ark_bulletproofs_amd as A

    O = oracle
    for cv in (0, 1):
        e = A.Engine(curve=cv)
        e.set_tuning(0, 1)
        e.set_tuning(1, 1)
        e.set_tuning(2, freeze)
        G, H, Q, a, b, Gf, Hf, P = _ipa_instance(O, cv, n)
        H = H.copy()
        H[0] = 0                      # an identity among the folded points (Z = 0 inside the shared inversion)
        tr = O.Transcript(b"innerproducttest")
        tr_ref = tr.clone()
        Lo, Ro, ao, bo = O.ipa_create(cv, tr_ref, Q, Gf, Hf, G, H, a, b)
        tr.append_message(b"dom-sep", b"ipp v1")
        tr.append_u64(b"n", n)
        Lg, Rg, ag, bg = e.ipa_create(Q, Gf, Hf, G, H, a, b, _challenger(O, cv, tr))
        assert (Lg == Lo).all() and (Rg == Ro).all() and (ag == ao).all() and (bg == bo).all()
        e.close()


def test_ipa_stepping_api_matches_callback_api(eng, oracle):
    """bp_ipa_begin / round_LR / round_fold / finish = bp_ipa_create, and bp_ipa_export returns the vectors of the round
    (with the pending generator factors) such that a fresh instance started from them finishes the same proof"""
    O, cv = oracle, eng.curve
    n = 32
    G, H, Q, a, b, Gf, Hf, P = _ipa_instance(O, cv, n)
    tr = O.Transcript(b"innerproducttest")
    Lo, Ro, ao, bo = O.ipa_create(cv, tr.clone(), Q, Gf, Hf, G, H, a, b)
    t1 = tr.clone()
    t1.append_message(b"dom-sep", b"ipp v1")
    t1.append_u64(b"n", n)
    ch = _challenger(O, cv, t1)
    eng.ipa_begin(Q, Gf, Hf, G, H, a, b)
    Ls, Rs = [], []
    for rnd in range(5):
        if rnd == 3:   # export after three folds, restart from the exported state (factors = the pending gammas)
            a4, b4, G4, H4, gG, gH = eng.ipa_export(n)
            assert len(a4) == 4
            eng.ipa_begin(Q, np.tile(gG, (4, 1)), np.tile(gH, (4, 1)), G4, H4, a4, b4)
        L, R = eng.ipa_round_LR()
        eng.ipa_round_fold(ch(L, R))
        Ls.append(L)
        Rs.append(R)
    ag, bg = eng.ipa_finish()
    assert (np.array(Ls) == Lo).all() and (np.array(Rs) == Ro).all() and (ag == ao).all() and (bg == bo).all()
    import ark_bulletproofs_amd as A

    with pytest.raises(A.ArkbpError):
        eng.ipa_round_LR()            # finished: no instance
    eng.ipa_begin(Q, Gf, Hf, G, H, a, b)
    with pytest.raises(A.ArkbpError):
        eng.ipa_round_fold(ao)        # fold before L, R
    with pytest.raises(A.ArkbpError):
        eng.ipa_finish()              # vectors still longer than 1


@pytest.mark.parametrize("world,n", [(2, 16), (4, 64), (4, 4), (2, 2)])
def test_ipa_index_cyclic_ranks_as_threads(oracle, world, n):
    """IPA with every vector partitioned i mod world.  The ranks run as threads of this process (one Engine each on
    the same GPU, an in-process all-gather); every rank must return the oracle's L_vec, R_vec, a, b."""
    import threading

    import ark_bulletproofs_amd as A
    from ark_bulletproofs_amd import engine as E
    from ark_bulletproofs_amd import parallel as P

    O = oracle
    for cv in (0, 1):
        G, H, Q, a, b, Gf, Hf, Pt = _ipa_instance(O, cv, n)
        u0 = O.fe_rand(O.fid(cv, True), bytes([9]) * 32, 1)[0]
        Gf = Gf.copy()
        Gf[n // 2:] = u0
        tr = O.Transcript(b"innerproducttest")
        Lo, Ro, ao, bo = O.ipa_create(cv, tr.clone(), Q, Gf, Hf, G, H, a, b)
        bar = threading.Barrier(world)
        slots = [None] * world
        results = [None] * world
        errors = []

        def run(rank):
            try:
                eng = A.Engine(curve=cv)
                t1 = tr.clone()
                t1.append_message(b"dom-sep", b"ipp v1")
                t1.append_u64(b"n", n)

                def allgather(arr):
                    slots[rank] = np.array(arr, copy=True)
                    bar.wait()
                    out = np.stack(slots)
                    bar.wait()
                    return out

                results[rank] = P.sharded_ipa_create(cv, eng, Q, Gf, Hf, G, H, a, b, _challenger(O, cv, t1), E.host_points_sum, rank, world, allgather=allgather)
                eng.close()
            except Exception as e:   # release the others
                errors.append(e)
                bar.abort()

        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        for r in range(world):
            L, R, ag, bg = results[r]
            assert (L == Lo).all() and (R == Ro).all() and (ag == ao).all() and (bg == bo).all()
