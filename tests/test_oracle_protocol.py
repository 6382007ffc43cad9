"""The reference's behavioural tests restated on the CPU oracle, both curves:
src/inner_product_proof.rs:411-553 (make_ipp_{1,2,4,32,64}), tests/r1cs_secq256k1.rs (shuffle 1..42,
example gadget +/-, serialization round trip, range proofs, batch_verify with mixed sizes and negatives)."""
import numpy as np
import pytest

SEED = bytes([7]) * 32


def _ipa_instance(O, cv, n):
    FR = O.fid(cv, True)
    G, H = O.bp_gens(cv, n)
    Q = O.scalar_mul(cv, O.generator(cv), O.fe_from_int(FR, 12345))
    a, b = O.fe_rand(FR, bytes([1]) * 32, n), O.fe_rand(FR, bytes([2]) * 32, n)
    yinv = O.fe_rand(FR, bytes([3]) * 32, 1)[0]
    Gf = np.tile(O.fe_from_int(FR, 1), (n, 1))
    Hf = np.zeros((n, 4), dtype=np.uint64)
    cur = O.fe_from_int(FR, 1)
    for i in range(n):
        Hf[i] = cur
        cur = O.fe_op("mul", FR, cur, yinv)
    bp = np.array([O.fe_op("mul", FR, b[i], Hf[i]) for i in range(n)])
    c = O.fe_from_int(FR, 0)
    for i in range(n):
        c = O.fe_op("add", FR, c, O.fe_op("mul", FR, a[i], b[i]))
    P = O.msm(cv, np.concatenate([G, H, Q.reshape(1, 8)]), np.concatenate([a, bp, c.reshape(1, 4)]))
    return G, H, Q, a, b, Gf, Hf, P


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 4, 32, 64])
def test_make_ipp(oracle, curve, n):
    O = oracle
    G, H, Q, a, b, Gf, Hf, P = _ipa_instance(O, curve, n)
    L, R, ao, bo = O.ipa_create(curve, O.Transcript(b"innerproducttest"), Q, Gf, Hf, G, H, a, b)
    assert len(L) == len(R) == n.bit_length() - 1
    assert O.ipa_verify(curve, O.Transcript(b"innerproducttest"), n, Gf, Hf, P, Q, G, H, L, R, ao, bo) == 0
    bad = O.fe_op("add", O.fid(curve, True), ao, O.fe_from_int(O.fid(curve, True), 1))
    assert O.ipa_verify(curve, O.Transcript(b"innerproducttest"), n, Gf, Hf, P, Q, G, H, L, R, bad, bo) != 0


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7, 24, 42])
def test_shuffle_gadget(oracle, curve, k):
    cap = max(1, 1 << (2 * k - 1).bit_length())
    pr = oracle.r1cs_prove(curve, oracle.SC_SHUFFLE, [k], SEED, cap)
    assert pr.rc == 0
    n = 0 if k == 1 else 2 * (k - 1)
    lg = max(n - 1, 0).bit_length() if n > 1 else 0
    assert len(pr.proof) == 539 + 66 * lg  # SURVEY.md §8 a16 wire size
    assert oracle.r1cs_verify(curve, oracle.SC_SHUFFLE, [k], cap, pr.proof, pr.commitments, pr.publics) == 0
    # tampered statement: swap two commitments of the input side
    if k >= 3:
        cm = pr.commitments.copy()
        cm[[0, k]] = cm[[k, 0]]
        assert oracle.r1cs_verify(curve, oracle.SC_SHUFFLE, [k], cap, pr.proof, cm, pr.publics) != 0


@pytest.mark.parametrize("curve", [0, 1])
def test_example_gadget_and_serialization(oracle, curve):
    O = oracle
    pr = O.r1cs_prove(curve, O.SC_EXAMPLE, [3, 4, 6, 1, 40, 9], SEED, 128)
    assert pr.rc == 0 and O.r1cs_verify(curve, O.SC_EXAMPLE, [3, 4, 6, 1, 40, 9], 128, pr.proof, pr.commitments, pr.publics) == 0
    pr = O.r1cs_prove(curve, O.SC_EXAMPLE, [3, 4, 6, 1, 40, 10], SEED, 128)
    assert pr.rc == 0 and O.r1cs_verify(curve, O.SC_EXAMPLE, [3, 4, 6, 1, 40, 10], 128, pr.proof, pr.commitments, pr.publics) != 0
    # malformed bytes -> FormatError (4)
    assert O.r1cs_verify(curve, O.SC_EXAMPLE, [3, 4, 6, 1, 40, 9], 128, pr.proof[:-1], pr.commitments, pr.publics) == 4
    bad = bytearray(pr.proof)
    bad[32] = 0xC0
    assert O.r1cs_verify(curve, O.SC_EXAMPLE, [3, 4, 6, 1, 40, 9], 128, bytes(bad), pr.commitments, pr.publics) == 4
    # too few generators -> InvalidGeneratorsLength (2) on the verifier side
    pr = O.r1cs_prove(curve, O.SC_RANGE, [32, 77], SEED, 128)
    assert pr.rc == 0


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("n", [2, 10, 32, 63])
def test_range_proof_gadget(oracle, curve, n):
    O = oracle
    mx = (1 << n) - 1
    for v in [0, mx // 3, mx]:
        pr = O.r1cs_prove(curve, O.SC_RANGE, [n, v], SEED, 128)
        assert pr.rc == 0 and O.r1cs_verify(curve, O.SC_RANGE, [n, v], 128, pr.proof, pr.commitments, pr.publics) == 0
    pr = O.r1cs_prove(curve, O.SC_RANGE, [n, mx + 1], SEED, 128)
    assert pr.rc == 0 and O.r1cs_verify(curve, O.SC_RANGE, [n, mx + 1], 128, pr.proof, pr.commitments, pr.publics) != 0


@pytest.mark.parametrize("curve", [0, 1])
def test_batch_range_proof_gadget(oracle, curve):
    O = oracle

    def run(vals):
        inst = []
        for i, (v, n) in enumerate(vals):
            pr = O.r1cs_prove(curve, O.SC_RANGE, [n, v], bytes([9 + i]) * 32, 128)
            assert pr.rc == 0
            inst.append((O.SC_RANGE, [n, v], pr.proof, pr.commitments, pr.publics))
        return O.batch_verify(curve, inst, 128, bytes([5]) * 32)

    assert run([(0, 16)]) == 0
    assert run([(0, 16), (3, 16), ((1 << 16) - 1, 16), (1 << 16, 32)]) == 0
    assert run([(0, 16), (3, 16), (1 << 16, 16), (1 << 16, 32)]) != 0
    assert run([(0, 16), (3, 16), ((1 << 16) - 1, 16), (1 << 16, 32), (1 << 63, 64)]) == 0
    assert run([(0, 16), (3, 16), ((1 << 16) - 1, 16), (1 << 32, 32), (1 << 63, 64)]) != 0


@pytest.mark.parametrize("curve", [0, 1])
def test_square_chain_and_multi_range(oracle, curve):
    O = oracle
    for sc, prm_ok, prm_bad in [(O.SC_SQUARE_CHAIN, [13, 0], [13, 1]), (O.SC_MULTI_RANGE, [3, 8, 0], [3, 8, 1])]:
        pr = O.r1cs_prove(curve, sc, prm_ok, SEED, 128, m_cap=16)
        assert pr.rc == 0 and O.r1cs_verify(curve, sc, prm_ok, 128, pr.proof, pr.commitments, pr.publics) == 0
        pr = O.r1cs_prove(curve, sc, prm_bad, SEED, 128, m_cap=16)
        assert pr.rc == 0 and O.r1cs_verify(curve, sc, prm_bad, 128, pr.proof, pr.commitments, pr.publics) != 0
    # generators too short -> InvalidGeneratorsLength (2)
    assert O.r1cs_prove(curve, O.SC_SQUARE_CHAIN, [13, 0], SEED, 128, m_cap=16).rc == 0


def test_verification_scalars_layout(oracle):
    O = oracle
    pr = O.r1cs_prove(0, O.SC_RANGE, [10, 5], SEED, 128)
    rc, sc = O.r1cs_verification_scalars(0, O.SC_RANGE, [10, 5], 128, pr.proof, pr.commitments, pr.publics, 4096)
    N, m, k = 16, 1, 4
    assert rc == 0 and len(sc) == 2 + 2 * N + 6 + m + 5 + 2 * k
