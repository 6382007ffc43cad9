"""CPU tests of the PRODUCT's host-side logic (no GPU): its own Merlin/STROBE, SHA3-512, ChaCha20-based
challenge scalars, PedersenGens::default and GeneratorsChain — against published vectors and against the
independent oracle."""
import hashlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def E():
    from ark_bulletproofs_amd import build

    build.build()
    from ark_bulletproofs_amd import engine

    return engine


def test_product_sha3(E):
    for n in [0, 1, 71, 72, 73, 500]:
        m = bytes((i * 5 + 1) & 255 for i in range(n))
        assert E.host_sha3_512(m) == hashlib.sha3_512(m).digest()


def test_product_merlin_vectors(E):
    t = E.HostTranscript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    t = E.HostTranscript(b"test protocol")
    t.append_message(b"step1", b"some data")
    data = bytes([99]) * 1024
    for _ in range(32):
        chl = t.challenge_bytes(b"challenge", 32)
        t.append_message(b"bigdata", data)
        t.append_message(b"challengedata", chl)
    assert chl.hex() == "a8c933f54fae76e3f9bea93648c1308e7dfa2152dd51674ff3ca438351cf003c"


@pytest.mark.parametrize("curve", [0, 1])
def test_product_transcript_protocol_matches_oracle(E, oracle, curve):
    O = oracle
    G, H = O.bp_gens(curve, 4)
    tp, to = E.HostTranscript(b"xyz"), O.Transcript(b"xyz")
    for i, p in enumerate(list(G) + [np.zeros(8, dtype=np.uint64)]):
        tp.append_point(curve, b"P", p)
        to.append_point(curve, b"P", p)
        assert (tp.challenge_scalar(curve, b"c") == to.challenge_scalar(curve, b"c")).all()


@pytest.mark.parametrize("curve", [0, 1])
def test_product_generators_match_oracle(E, oracle, curve):
    O = oracle
    B, Bb = E.pedersen_gens(curve)
    Bo, Bbo = O.pedersen_default(curve)
    assert (B == Bo).all() and (Bb == Bbo).all()
    n = 300  # > 256 attempts: exercises the threaded path
    Go, Ho = O.bp_gens(curve, n)
    assert (E.host_derive_generators(curve, 0, 0, n) == Go).all()
    assert (E.host_derive_generators(curve, 1, 0, n) == Ho).all()
    g1, _ = O.bp_gens_party(curve, 8, 3)
    assert (E.host_derive_generators(curve, 0, 3, 8) == g1).all()


@pytest.mark.parametrize("curve", [0, 1])
def test_product_transcript_rng_scalar_and_x8(E, oracle, curve):
    """The prover's blinding stream: product scalar path (with the collapsed steady-state STROBE step) and the AVX-512
    Keccak-f x8 lockstep path must both equal the oracle's plain merlin TranscriptRng, draw for draw."""
    O = oracle
    FR = O.fid(curve, True)
    wit = O.fe_rand(FR, bytes([77]) * 32, 3)
    count = 40
    tp, to = E.HostTranscript(b"rngtest"), O.Transcript(b"rngtest")
    tp.append_message(b"x", b"abc")
    to.append_message(b"x", b"abc")
    seeds = [bytes([10 + j]) * 32 for j in range(8)]
    ref = [O.Transcript(handle=None, label=b"unused") if False else None for _ in range(8)]
    exp = np.stack([to.rng_draws(curve, wit, seeds[j], count) for j in range(8)])
    got1 = E.debug_rng_draws(curve, tp, wit, seeds[0], count)
    assert (got1[0] == exp[0]).all()
    got8 = E.debug_rng_draws(curve, tp, wit, b"".join(seeds), count)
    if got8 is None:
        pytest.skip("no AVX-512 on this host: the x8 path is not taken")
    assert (got8 == exp).all()


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("lanes,npts,prefix", [(8, 256, 0), (8, 5, 3), (3, 40, 77), (2, 1, 165)])
def test_product_lockstep_commitment_appends_equal_scalar_appends(E, oracle, curve, lanes, npts, prefix):
    """batch verification appends the commitments of eight same-shaped instances in lockstep (host::StrobeX8): every transcript
    must end in exactly the state the one-at-a-time appends leave (checked through a challenge and further use), for any STROBE
    position the prefix leaves (here 0 .. 165 of the 166-byte rate)."""
    O = oracle
    G, H = O.bp_gens(curve, max(npts, 2) * 4)
    pool = np.concatenate([G, H, np.zeros((1, 8), dtype=np.uint64)])   # (the identity serialises differently: flag 0x40)
    pts = np.stack([[pool[(l * 31 + v * 7 + (v * v) % 5) % len(pool)] for v in range(npts)] for l in range(lanes)])
    a = [E.HostTranscript(b"lockstep") for _ in range(lanes)]
    b = [E.HostTranscript(b"lockstep") for _ in range(lanes)]
    for t in a + b:
        t.append_message(b"pad", bytes(range(prefix % 256)) * 1 if prefix else b"")
    if not E.debug_append_points_x8(curve, a, b"V", pts):
        pytest.skip("no AVX-512 on this host: the product takes the scalar path")
    for l in range(lanes):
        for v in range(npts):
            b[l].append_point(curve, b"V", pts[l, v])
    for l in range(lanes):
        assert a[l].challenge_bytes(b"c", 64) == b[l].challenge_bytes(b"c", 64)
        a[l].append_message(b"more", b"x" * 200)
        b[l].append_message(b"more", b"x" * 200)
        assert a[l].challenge_bytes(b"d", 16) == b[l].challenge_bytes(b"d", 16)
    assert len({a[l].challenge_bytes(b"e", 8) for l in range(lanes)}) == lanes or npts == 0


def test_product_glv_decomposition():
    """Host-side GLV split used by the uniform IPA fold on secq256k1: t = t1 + lambda*t2 (mod r), both halves given as
    non-adjacent signed digits of at most 130 positions; lambda is a primitive cube root of unity in Fr."""
    import ctypes as C
    import random

    from ark_bulletproofs_amd import _lib

    L = _lib.lib()
    r = 2**256 - 2**32 - 977   # secq256k1 scalar field = secp256k1 base field
    R = pow(2, 256, r)
    rnd = random.Random(7)
    samples = [0, 1, 2, r - 1, r - 2, (r - 1) // 2, 2**128, 2**129 - 1] + [rnd.randrange(r) for _ in range(300)]
    lam = None
    for t in samples:
        tm = np.array([((t * R % r) >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
        masks = np.zeros(20, dtype=np.uint32)
        lam_out = np.zeros(4, dtype=np.uint64)
        rc = L.bp_debug_glv_decompose(0, _lib.ptr(tm), _lib.ptr(masks), _lib.ptr(lam_out))
        assert rc == 0
        lam_t = sum(int(lam_out[i]) << (64 * i) for i in range(4))
        if lam is None:
            lam = lam_t
            assert lam != 1 and pow(lam, 3, r) == 1
        assert lam_t == lam

        def val(plus, minus):
            p = sum(int(plus[i]) << (32 * i) for i in range(5))
            m = sum(int(minus[i]) << (32 * i) for i in range(5))
            assert p & m == 0 and (p | m) < 2**130
            nz = p | m
            assert nz & (nz >> 1) == 0   # non-adjacent
            return p - m

        t1 = val(masks[0:5], masks[5:10])
        t2 = val(masks[10:15], masks[15:20])
        assert abs(t1) < 2**129 and abs(t2) < 2**129
        assert (t1 + lam * t2 - t) % r == 0
    masks = np.zeros(20, dtype=np.uint32)
    assert L.bp_debug_glv_decompose(1, _lib.ptr(tm), _lib.ptr(masks), _lib.ptr(lam_out)) != 0   # zorro: no endomorphism


@pytest.mark.parametrize("curve", [0, 1])
def test_product_transcript_rng_matches_independent_model(E, curve):
    """the PRODUCT's TranscriptRng (csrc/host_proto.hpp, scalar path and the AVX-512 x8 lockstep path) against the independent
    STROBE model of tests/pystrobe.py — no oracle involved"""
    import pystrobe as PS

    p = [0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F, (1 << 255) - 19][curve]
    R = 1 << 256
    wit_int = [7, 1 << 200, p - 2]
    witness = np.array([[(v * R % p >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in wit_int], dtype=np.uint64)

    def model(seed, count):
        tm = PS.Transcript(b"rng pin")
        b = tm.build_rng()
        for v in wit_int:
            b.rekey_with_witness_bytes(b"v_blinding", v.to_bytes(32, "little"))
        rng = b.finalize(PS.chacha20_block(seed, 0)[:32])
        out = []
        while len(out) < count:
            limbs = [rng.next_u64() for _ in range(4)]
            limbs[3] &= (1 << (64 - (256 - p.bit_length()))) - 1
            v = sum(l << (64 * i) for i, l in enumerate(limbs))
            if v < p:
                out.append(v)
        return out

    def ints(rows):
        return [sum(int(r[i]) << (64 * i) for i in range(4)) for r in rows]

    seed = bytes([33]) * 32
    got = E.debug_rng_draws(curve, E.HostTranscript(b"rng pin"), witness, seed, 5)
    assert ints(got[0]) == model(seed, 5)
    seeds8 = b"".join(bytes([40 + j]) * 32 for j in range(8))
    got8 = E.debug_rng_draws(curve, E.HostTranscript(b"rng pin"), witness, seeds8, 12)
    if got8 is not None:      # AVX-512 hosts: the x8 stream yields raw (unrejected) limb groups; every accepted one must be the model's next value
        for j in range(8):
            exp = model(seeds8[32 * j: 32 * j + 32], 12)
            vals = [v for v in ints(got8[j]) if v < p]
            assert vals[: len(vals)] == exp[: len(vals)] and len(vals) >= 10


def test_bench_cfg5_leg_cannot_cost_the_run_its_json_line():
    """bench.py's multi-GPU headline run adds the partitioned 2^22 proof as a third leg whose collectives no one-GPU box can
    exercise.  If that leg hangs (a rank failed alone, a collective never completes) or raises, rank 0 must still print the
    headline line — prove and verify are complete by then — with the failure recorded under "cfg5"."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = r"""
import sys, time, json, argparse
sys.path.insert(0, %r)
import bench
mode = sys.argv[1]
def fake(args, rank, world, local):
    if mode == "hang":
        time.sleep(60)
    raise RuntimeError("rank failed in the cfg5 leg")
bench.run_cfg5 = fake
args = argparse.Namespace(cfg5_timeout=1, strict_exit=(sys.argv[2] == "strict"))
res = {"metric": "m", "value": 1.0, "verify": {"value": 2.0}}
res["cfg5"] = bench.guarded_cfg5(args, 0, 2, 0, res)
print(json.dumps(res), flush=True)
""" % root
    for mode, strict in (("hang", "lenient"), ("raise", "lenient"), ("hang", "strict")):
        out = subprocess.run([sys.executable, "-c", prog, mode, strict], capture_output=True, text=True, timeout=50)
        # the line always comes; --strict-exit also reports the failed leg through the exit status
        assert out.returncode == (3 if strict == "strict" else 0), out.stderr[-500:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout
        d = json.loads(lines[0])
        assert d["value"] == 1.0 and d["verify"]["value"] == 2.0 and "error" in d["cfg5"], d


@pytest.mark.parametrize("prefix,msg_len,nbytes", [(0, 8, 32), (3, 65, 32), (150, 11, 32), (165, 6, 64), (77, 0, 1), (140, 40, 33)])
def test_product_lockstep_challenges_equal_scalar_challenges(E, prefix, msg_len, nbytes):
    """the batch verifier runs the whole Fiat-Shamir replay of eight same-shaped proofs in lockstep (host::StrobeX8: gather, appends
    with the same or with per-lane messages, challenge_bytes_each, scatter): challenges and final states must equal eight ordinary
    transcripts', for any STROBE position (0 .. 165 of the 166-byte rate) and for challenge lengths that cross 64-bit words and the
    rate boundary."""
    a = [E.HostTranscript(b"lockstep-challenge") for _ in range(8)]
    b = [E.HostTranscript(b"lockstep-challenge") for _ in range(8)]
    for l in range(8):
        for t in (a[l], b[l]):
            t.append_message(b"lane", bytes([l]) * 8)                  # different contents, same position
            if prefix:
                t.append_message(b"pad", bytes(range(prefix)))
    msg = bytes((7 * i + 1) % 256 for i in range(msg_len))
    got = E.debug_challenge_x8(a, b"m", msg, b"y", nbytes)
    if got is None:
        pytest.skip("no AVX-512 on this host: the product takes the scalar path")
    for l in range(8):
        b[l].append_message(b"m", msg)
        assert bytes(got[l]) == b[l].challenge_bytes(b"y", nbytes), l
        a[l].append_message(b"after", b"z" * 170)
        b[l].append_message(b"after", b"z" * 170)
        assert a[l].challenge_bytes(b"c", 40) == b[l].challenge_bytes(b"c", 40), l
    assert len({bytes(got[l]) for l in range(8)}) == 8


def test_lockstep_replay_schedule_equals_live_transcript(oracle):
    """replay_challenges_x8 re-states verify_prepare_t's Fiat-Shamir schedule (labels, the "m" / "n" encodings, the
    1-phase separator, message order) for eight transcripts in lockstep; any edit to one copy would silently desynchronise the
    challenges.  CPU-only comparison of both on proofs from the oracle's prover, both curves; a group with a differing round count
    and a group with a randomized (two-phase) statement must be declined, not mis-replayed."""
    from ark_bulletproofs_amd import engine as E

    O = oracle
    for cv in (0, 1):
        inst = []
        for j in range(8):
            pr = O.r1cs_prove(cv, O.SC_MULTI_RANGE, [3, 8, 0], bytes([30 + j]) * 32, 64, m_cap=8)
            assert pr.rc == 0
            inst.append((O.SC_MULTI_RANGE, [3, 8, 0], pr.proof, pr.commitments, pr.publics))
        live = E.debug_verify_challenges(cv, inst, False)
        x8 = E.debug_verify_challenges(cv, inst, True)
        assert all(len(c) == 6 + 5 for c in live)          # y z u x w, u_1..u_5 (n = 24 -> 32), r
        if x8 is None:
            pytest.skip("no AVX-512 on this host: the lockstep replay is not used")
        for a, b in zip(live, x8):
            assert (a == b).all(), "lockstep replay and live transcript disagree"
        # a lane with another round count: declined
        pr = O.r1cs_prove(cv, O.SC_MULTI_RANGE, [3, 4, 0], bytes([77]) * 32, 64, m_cap=8)
        odd = list(inst); odd[5] = (O.SC_MULTI_RANGE, [3, 4, 0], pr.proof, pr.commitments, pr.publics)
        assert E.debug_verify_challenges(cv, odd, True) is None
        assert len(E.debug_verify_challenges(cv, odd, False)[5]) == 6 + 4
        # a two-phase lane (the shuffle gadget's randomized constraints): declined
        pr = O.r1cs_prove(cv, O.SC_SHUFFLE, [4], bytes([78]) * 32, 64, m_cap=16)
        odd = list(inst); odd[2] = (O.SC_SHUFFLE, [4], pr.proof, pr.commitments, pr.publics)
        assert E.debug_verify_challenges(cv, odd, True) is None


def test_direct_table_round_visits_exactly_the_bases_with_nonzero_scalars():
    """The small-statement inner-product argument (csrc/small.cuh) never folds G and H: in the round with half length n, generator t of
    the n0 original ones carries a scalar for L iff its position in the current vector, t mod 2n, lies in the UPPER half (G) / LOWER half
    (H), and for R the other way round (k_ipa_frozen_scalars, src/inner_product_proof.rs:92-102, 112-122).  The table sums visit only
    those: term j of a run with `fold_n = n` stands for element (j // n) * 2n + j % n + (n if fold_hi else 0) (DtSeg in small.cuh).
    This is the index algebra of that mapping, for every round of a few sizes: each job enumerates its half exactly once, L and R
    partition the generators, and the pairing a[(pos + n) mod 2n] <-> G[t] is the reference's (a_L with G_R, a_R with G_L)."""
    for lg in range(1, 9):
        n0 = 1 << lg
        n = n0 // 2
        while n >= 1:
            def run(fold_hi):
                return [(j // n) * 2 * n + j % n + (n if fold_hi else 0) for j in range(n0 // 2)]
            upper, lower = run(True), run(False)
            assert sorted(upper) == [t for t in range(n0) if t % (2 * n) >= n]
            assert sorted(lower) == [t for t in range(n0) if t % (2 * n) < n]
            assert len(set(upper) | set(lower)) == n0 and not (set(upper) & set(lower))
            for t in range(n0):
                pos = t % (2 * n)
                partner = pos + n if pos < n else pos - n        # k_dt_round: idx
                assert partner == (pos + n) % (2 * n)
                # L: G at upper positions pairs with a_L[pos - n]; R: G at lower positions pairs with a_R[pos] = a[pos + n]
                if pos >= n:
                    assert partner == pos - n and partner < n
                else:
                    assert partner == pos + n and partner >= n
            n //= 2
