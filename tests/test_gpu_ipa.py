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
