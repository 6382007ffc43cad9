"""The verifier front end on the device (csrc/vfe.hip; include/arkbp.h "verifier front end on the device"): for batches of
like-instances of one single-phase statement the wire codec, the merlin transcript replay (Keccak-f / STROBE / ChaCha20 ->
Fr::rand) and the O(k + m) challenge arithmetic of `batch_verify` (src/r1cs/verifier.rs:604-691, :403-541;
src/inner_product_proof.rs:244-314; src/transcript.rs:45-102) run as GPU kernels.  Checked here: every device-derived
challenge against the product's host transcript AND the oracle's; accept / reject and the mega-check POINT against the host
replay (BP_TUNE_VFY_DEVICE = 0) and the oracle for valid and failing batches; inputs the reference rejects take the host replay
and come back with the reference's error; both entry points (scenario batches, recorded handles with like-instances)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
OK, E_VERIFICATION, E_FORMAT = 0, -4, -6
SC_SHUFFLE, SC_RANGE, SC_EXAMPLE, SC_SQUARE_CHAIN, SC_MULTI_RANGE = 0, 1, 2, 3, 4
LABELS = {SC_RANGE: b"RangeProofTest", SC_SQUARE_CHAIN: b"SquareChainBenchmark", SC_MULTI_RANGE: b"MultiRangeBenchmark"}   # host_proto.hpp scenario_label
TUNE_VFY_DEVICE = 11


@pytest.fixture(scope="module", params=[0, 1], ids=["secq256k1", "zorro"])
def eng(request):
    import ark_bulletproofs_amd as A

    e = A.Engine(curve=request.param)
    e.gens_derive(256)
    yield e
    e.close()


def scenario_label(E, sc):
    return LABELS[sc]


def parse_proof(O, cv, proof):
    """wire order: 11 points, 3 scalars, L_vec, R_vec, a, b (src/r1cs/proof.rs:74-81)"""
    k = (len(proof) - 539) // 66
    pts = [O.point_deser_compressed(cv, proof[33 * j: 33 * j + 33]) for j in range(11)]
    sc = [proof[363 + 32 * j: 395 + 32 * j] for j in range(3)]
    L = [O.point_deser_compressed(cv, proof[467 + 33 * i: 500 + 33 * i]) for i in range(k)]
    R = [O.point_deser_compressed(cv, proof[475 + 33 * k + 33 * i: 508 + 33 * k + 33 * i]) for i in range(k)]
    return k, pts, sc, L, R


def replay(T, cv, m, k, V, pts, sc, L, R, point, message, challenge):
    """verify_prepare_t's schedule (src/r1cs/verifier.rs:403-460, 516-519) on a live transcript T"""
    out = []
    for v in V:
        point(T, b"V", v)
    message(T, b"m", int(m).to_bytes(8, "little"))
    for lab, p in zip((b"A_I1", b"A_O1", b"S1"), pts[:3]):
        point(T, lab, p)
    message(T, b"dom-sep", b"r1cs-1phase")
    for lab, p in zip((b"A_I2", b"A_O2", b"S2"), pts[3:6]):
        point(T, lab, p)
    out += [challenge(T, b"y"), challenge(T, b"z")]
    for lab, p in zip((b"T_1", b"T_3", b"T_4", b"T_5", b"T_6"), pts[6:11]):
        point(T, lab, p)
    out += [challenge(T, b"u"), challenge(T, b"x")]
    for lab, s in zip((b"t_x", b"t_x_blinding", b"e_blinding"), sc):
        message(T, lab, s)
    out.append(challenge(T, b"w"))
    message(T, b"dom-sep", b"ipp v1")
    message(T, b"n", int(1 << k).to_bytes(8, "little"))
    for i in range(k):
        point(T, b"L", L[i]); point(T, b"R", R[i])
        out.append(challenge(T, b"u"))
    out.append(challenge(T, b"r"))
    return out


@pytest.mark.parametrize("sc,prm", [(SC_MULTI_RANGE, [3, 8, 0]), (SC_SQUARE_CHAIN, [13, 0]), (SC_RANGE, [32, 77])])
def test_device_challenges_equal_host_transcript_and_oracle(eng, oracle, sc, prm):
    from ark_bulletproofs_amd import engine as E

    O, cv = oracle, eng.curve
    proofs, Vs = [], []
    for i in range(70):   # more than one wavefront of the sponge kernel
        pr = eng.prove_scenario(sc, prm, bytes([3 + (i % 5)]) * 32, m_cap=16) if i < 5 else None
        if pr is not None:
            proofs.append(pr.proof); Vs.append(np.asarray(pr.commitments, dtype=np.uint64).reshape(-1, 8))
        else:
            proofs.append(proofs[i % 5]); Vs.append(Vs[i % 5])
    m = len(Vs[0])
    t0 = E.HostTranscript(scenario_label(E, sc))
    t0.append_message(b"dom-sep", b"r1cs v1")
    state = E.transcript_state(t0)
    seeds, chal, status = eng.debug_vfe_challenges(proofs, np.stack(Vs), state, True)
    assert status == 0
    for i in (0, 1, 4, 63, 64, 69):
        k, pts, scal, L, R = parse_proof(O, cv, proofs[i])
        # the product's host transcript
        th = E.transcript_from_state(state)
        exp_h = replay(th, cv, m, k, Vs[i], pts, scal, L, R, lambda T, l, p: T.append_point(cv, l, p), lambda T, l, b: T.append_message(l, b),
                       lambda T, l: T.challenge_scalar(cv, l))
        assert (np.stack(exp_h) == chal[i]).all(), "device challenges differ from the host transcript (proof %d)" % i
        # the oracle's transcript, from the label on
        to = O.Transcript(scenario_label(E, sc))
        to.append_message(b"dom-sep", b"r1cs v1")
        exp_o = replay(to, cv, m, k, Vs[i], pts, scal, L, R, lambda T, l, p: T.append_point(cv, l, p), lambda T, l, b: T.append_message(l, b),
                       lambda T, l: T.challenge_scalar(cv, l))
        assert (np.stack(exp_o) == chal[i]).all(), "device challenges differ from the oracle's transcript (proof %d)" % i
        # the squeezed 32-byte seeds against the CPU run of the same schedule over the oracle's serializations
        items = np.zeros((m + 11 + 2 * k + 3, 72), dtype=np.uint8)
        for j, v in enumerate(list(Vs[i]) + pts + L + R):
            items[j, :65] = np.frombuffer(O.point_ser(cv, v, False), dtype=np.uint8)
        for j, s in enumerate(scal):
            items[m + 11 + 2 * k + j, :32] = np.frombuffer(s, dtype=np.uint8)
        cpu_seeds, _ = E.vfe_schedule_replay(state, True, m, k, 1 << k, items)
        assert (cpu_seeds == seeds[i]).all()
    # transcripts that already hold the commitments (the recorded-handle form), one state per proof
    states = []
    for i in range(len(proofs)):
        t = E.transcript_from_state(state)
        for v in Vs[i]:
            t.append_point(cv, b"V", v)
        states.append(E.transcript_state(t))
    seeds2, chal2, status2 = eng.debug_vfe_challenges(proofs, np.stack(Vs), states, False)
    assert status2 == 0 and (chal2 == chal).all() and (seeds2 == seeds).all()


def test_device_flags_what_the_reference_rejects(eng, oracle):
    from ark_bulletproofs_amd import engine as E

    O, cv = oracle, eng.curve
    sc, prm = SC_MULTI_RANGE, [2, 8, 0]
    pr = eng.prove_scenario(sc, prm, bytes([9]) * 32, m_cap=16)
    V = np.asarray(pr.commitments, dtype=np.uint64).reshape(1, -1, 8)
    t0 = E.HostTranscript(scenario_label(E, sc))
    t0.append_message(b"dom-sep", b"r1cs v1")
    state = E.transcript_state(t0)
    k = (len(pr.proof) - 539) // 66

    def status_of(proof):
        return eng.debug_vfe_challenges([bytes(proof)], V, state, True)[2]

    assert status_of(pr.proof) == 0
    bad = bytearray(pr.proof); bad[32] = 0xC0            # both flag bits
    assert status_of(bad) & 1
    bad = bytearray(pr.proof); bad[32] |= 0x01           # an unknown flag bit
    assert status_of(bad) & 1
    bad = bytearray(pr.proof); bad[0:32] = b"\xff" * 32  # x >= p
    assert status_of(bad) & 1
    bad = bytearray(pr.proof); bad[363 + 31] = 0xff; bad[363:363 + 31] = b"\xff" * 31   # t_x >= r
    assert status_of(bad) & 1
    bad = bytearray(pr.proof); bad[6 * 33: 7 * 33] = b"\x00" * 32 + b"\x40"   # T_1 = identity: validate_and_append_point
    assert status_of(bad) & 2
    bad = bytearray(pr.proof); bad[3 * 33: 4 * 33] = b"\x00" * 32 + b"\x40"   # A_I2 = identity is what a single-phase proof carries
    assert status_of(bad) == 0
    bad = bytearray(pr.proof); bad[459] ^= 1             # L_vec announces another length
    assert status_of(bad) & 4
    # an x with no point on the curve: find one by stepping x
    bad = bytearray(pr.proof)
    for step in range(1, 64):
        bad[0] = (pr.proof[0] + step) & 0xff
        if O.point_deser_compressed(cv, bytes(bad[:33])) is None:
            break
    assert O.point_deser_compressed(cv, bytes(bad[:33])) is None
    assert status_of(bad) & 1


def make_batch(eng, sc, prm, count, distinct, m_cap=64):
    base = [eng.prove_scenario(sc, prm, bytes([50 + i]) * 32, m_cap=m_cap) for i in range(distinct)]
    return [(sc, prm, base[i % distinct].proof, base[i % distinct].commitments, base[i % distinct].publics) for i in range(count)]


# (the square chain's public output depends on the witness: distinct witnesses are distinct statements, i.e. not like-instances)
@pytest.mark.parametrize("sc,prm,count,distinct", [(SC_MULTI_RANGE, [5, 8, 0], 600, 3), (SC_SQUARE_CHAIN, [100, 0], 70, 1), (SC_RANGE, [16, 1234], 3, 3)])
def test_device_front_end_equals_host_replay_and_oracle(eng, oracle, sc, prm, count, distinct):
    """the same batch through the device front end and through the host replay: statuses and mega-check points equal, and equal
    to the oracle's MSM, for a valid batch and for batches with a wrong proof (point != identity, the SAME point)"""
    O, cv = oracle, eng.curve
    seed = bytes([5]) * 32
    inst = make_batch(eng, sc, prm, count, distinct)

    def both(instances):
        eng.set_tuning(TUNE_VFY_DEVICE, 1)
        d0, f0 = eng.vfe_stats()
        rc_d, _, pt_d = eng.batch_verify(instances, seed, want_point=True)
        d1, f1 = eng.vfe_stats()
        eng.set_tuning(TUNE_VFY_DEVICE, 0)
        rc_h, _, pt_h = eng.batch_verify(instances, seed, want_point=True)
        d2, f2 = eng.vfe_stats()
        eng.set_tuning(TUNE_VFY_DEVICE, 1)
        assert (d2, f2) == (d1, f1), "the host replay was asked for"
        return rc_d, pt_d, rc_h, pt_h, (d1 - d0, f1 - f0)

    rc_d, pt_d, rc_h, pt_h, used = both(inst)
    assert used == (1, 0), "the device front end did not take a batch of like-instances"
    assert rc_d == OK and rc_h == OK and not pt_d.any() and not pt_h.any()
    # a wrong scalar in one proof (well-formed): VerificationError with the same non-identity point everywhere
    bad_at = count - 2
    s, p, proof, cm, pb = inst[bad_at]
    bad = bytearray(proof); bad[11 * 33 + 40] ^= 2
    inst_bad = list(inst); inst_bad[bad_at] = (s, p, bytes(bad), cm, pb)
    rc_d, pt_d, rc_h, pt_h, used = both(inst_bad)
    assert used == (1, 0)
    assert rc_d == E_VERIFICATION and rc_h == E_VERIFICATION and pt_d.any() and (pt_d == pt_h).all()
    if count <= 70:
        rc_o, pt_o = O.batch_verify_point(cv, inst_bad, 256, seed)
        assert rc_o != 0 and (np.asarray(pt_o, dtype=np.uint64).reshape(-1) == pt_d).all(), "mega-check point differs from the oracle's MSM"
    # a wrong commitment
    cm2 = np.array(inst[1][3], dtype=np.uint64).reshape(-1, 8).copy()
    cm2[0] = O.generator(cv)
    inst_bad = list(inst); inst_bad[1] = (inst[1][0], inst[1][1], inst[1][2], cm2, inst[1][4])
    rc_d, pt_d, rc_h, pt_h, used = both(inst_bad)
    assert used == (1, 0) and rc_d == E_VERIFICATION and rc_h == E_VERIFICATION and (pt_d == pt_h).all()


def test_rejected_inputs_take_the_host_replay(eng, oracle):
    """a malformed or identity-carrying proof in the batch: the device front end flags it, the host replay reports the reference's
    error (FormatError before VerificationError in instance order as R1CSProof::from_bytes runs before batch_verify)"""
    O, cv = oracle, eng.curve
    seed = bytes([6]) * 32
    inst = make_batch(eng, SC_MULTI_RANGE, [2, 8, 0], 20, 2)
    d0, f0 = eng.vfe_stats()
    s, p, proof, cm, pb = inst[7]
    bad = bytearray(proof); bad[6 * 33: 7 * 33] = b"\x00" * 32 + b"\x40"      # T_1 = identity
    bad_inst = list(inst); bad_inst[7] = (s, p, bytes(bad), cm, pb)
    rc, _ = eng.batch_verify(bad_inst, seed)
    assert rc == E_VERIFICATION and O.batch_verify(cv, bad_inst, 256, seed) != 0
    bad2 = bytearray(proof); bad2[32] = 0xC0
    bad_inst[12] = (s, p, bytes(bad2), cm, pb)
    rc, _ = eng.batch_verify(bad_inst, seed)
    assert rc == E_FORMAT
    d1, f1 = eng.vfe_stats()
    assert (d1 - d0, f1 - f0) == (0, 2)
    # shapes the front end does not cover go to the host replay without being counted as fallbacks: a mixed batch ...
    mixed = inst[:5] + make_batch(eng, SC_RANGE, [16, 99], 3, 1)
    rc, _ = eng.batch_verify(mixed, seed)
    assert rc == OK
    # ... and a two-phase statement (the shuffle's randomized constraints append what its challenge decides)
    sh = make_batch(eng, SC_SHUFFLE, [4], 6, 2)
    rc, _ = eng.batch_verify(sh, seed)
    assert rc == OK
    assert eng.vfe_stats() == (d1, f1)


@pytest.mark.parametrize("curve", [0, 1])
def test_recorded_handles_with_like_instances_use_the_device_front_end(oracle, curve):
    """bp_r1cs_batch_verify over one recorded verifier and its like-instances (transcripts that already hold their commitments):
    same result and point as the host replay and as the oracle's batch_verify over per-instance recordings"""
    from ark_bulletproofs_amd import engine as E
    import gadgets as GD
    import test_gpu_cs_api as T

    O = oracle
    eng = E.Engine(curve=curve, device=0)
    eng.gens_derive(T.GENS)
    try:
        F = GD.Field(O, curve)
        kw = dict(n_mul=11, n_extra=3, n_alloc=1)
        struct_seed, m, count = 41, 3, 9
        items = [T.product_prove(E, eng, F, struct_seed, 80 + w, m, False, **kw) for w in range(3)]
        pubs = items[0][2]
        alphas = O.fe_rand(O.fid(curve, True), bytes([4]) * 32, count)

        def build(bad=None):
            # every instance shares instance 0's public constants (like-instances share the recording, constants included)
            v0 = T.product_verifier(E, curve, F, struct_seed, items[0][1], pubs, False, **kw)
            vs, ov, pf = [v0], [T.oracle_verifier(O, curve, F, struct_seed, items[0][1], pubs, False, **kw)], [items[0][0]]
            for i in range(1, count):
                vs.append(T.product_verifier(E, curve, F, struct_seed, items[0][1], pubs, False, like=v0, **kw))
                ov.append(T.oracle_verifier(O, curve, F, struct_seed, items[0][1], pubs, False, **kw))
                pf.append(items[0][0])
            if bad is not None:
                b = bytearray(pf[bad]); b[-40] ^= 2; pf[bad] = bytes(b)
            return vs, ov, pf

        for bad in (None, 6):
            vs, ov, pf = build(bad)
            d0, f0 = eng.vfe_stats()
            rc_d, pt_d = E.batch_verify_cs(eng, vs, pf, alphas, want_point=True)
            assert eng.vfe_stats() == (d0 + 1, f0)
            vs, _, _ = build(bad)
            eng.set_tuning(TUNE_VFY_DEVICE, 0)
            rc_h, pt_h = E.batch_verify_cs(eng, vs, pf, alphas, want_point=True)
            eng.set_tuning(TUNE_VFY_DEVICE, 1)
            rc_o, pt_o = O.batch_verify_cs(curve, ov, pf, T.GENS, alphas)
            assert rc_d == rc_h == (0 if bad is None else E_VERIFICATION)
            assert (rc_o == 0) == (bad is None)
            assert (pt_d == pt_h).all() and (pt_d == np.asarray(pt_o).reshape(-1)).all()
    finally:
        eng.close()
