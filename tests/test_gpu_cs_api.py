"""The caller's-own-gadget boundary (include/arkbp.h "bp_cs": Prover / Verifier / ConstraintSystem of the reference, src/r1cs/
constraint_system.rs:19-135, prover.rs:96-268 + 444-831, verifier.rs:69-224 + 549-691) against the oracle's restated Prover /
Verifier running the SAME gadget: proofs byte-identical, accept / reject identical, the batch mega-check point identical.
The gadgets (tests/gadgets.py) are random sparse constraint systems, 1- and 2-phase — not the product's scenarios."""
import numpy as np
import pytest

import gadgets as GD

pytestmark = pytest.mark.gpu
LABEL = b"GenericGadgetTest"
GENS = 256


@pytest.fixture(scope="module")
def E():
    from ark_bulletproofs_amd import engine

    return engine


@pytest.fixture(scope="module")
def engines(E):
    out = {}
    for cv in (0, 1):
        e = E.Engine(curve=cv, device=0)
        e.gens_derive(GENS)
        out[cv] = e
    yield out
    for e in out.values():
        e.close()


def product_prove(E, eng, F, struct_seed, wit_seed, m, two_phase, gpu_commit=False, **kw):
    vals, blinds = GD.make_witness(F, wit_seed, m)
    t = E.HostTranscript(LABEL)
    t.append_message(b"dom-sep", b"generic gadget v1")
    p = E.ProverCS(eng.curve, t)
    V, vars_ = p.commit([F.w(v) for v in vals], [F.w(b) for b in blinds], engine=eng if gpu_commit else None)
    wit = GD.Witness(F)
    for var, v in zip(vars_, vals):
        wit.val[var] = v
    publics = []
    GD.random_program(p, F, struct_seed, wit, vars_, two_phase=two_phase, publics=publics, **kw)
    proof = p.prove(eng, bytes([wit_seed & 255]) * 32)
    return proof, V, publics


def oracle_prove(O, curve, F, struct_seed, wit_seed, m, two_phase, **kw):
    vals, blinds = GD.make_witness(F, wit_seed, m)
    p = O.ProverCS(curve, LABEL)
    p.transcript().append_message(b"dom-sep", b"generic gadget v1")
    p.start()
    V, vars_ = p.commit([F.w(v) for v in vals], [F.w(b) for b in blinds])
    wit = GD.Witness(F)
    for var, v in zip(vars_, vals):
        wit.val[var] = v
    publics = []
    GD.random_program(p, F, struct_seed, wit, vars_, two_phase=two_phase, publics=publics, **kw)
    return p.prove(GENS, bytes([wit_seed & 255]) * 32), V, publics


def product_verifier(E, curve, F, struct_seed, V, publics, two_phase, like=None, **kw):
    t = E.HostTranscript(LABEL)
    t.append_message(b"dom-sep", b"generic gadget v1")
    v = E.VerifierCS(curve, t, like=like)
    vars_ = v.commit(V)
    if like is None:
        GD.random_program(v, F, struct_seed, None, vars_, two_phase=two_phase, publics=publics, **kw)
    return v


def oracle_verifier(O, curve, F, struct_seed, V, publics, two_phase, **kw):
    v = O.VerifierCS(curve, LABEL)
    v.transcript().append_message(b"dom-sep", b"generic gadget v1")
    v.start()
    vars_ = v.commit(V)
    GD.random_program(v, F, struct_seed, None, vars_, two_phase=two_phase, publics=publics, **kw)
    return v


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("two_phase", [False, True])
@pytest.mark.parametrize("struct_seed", [11, 12])
def test_generic_gadget_prove_is_byte_identical_and_verifies(E, engines, oracle, curve, two_phase, struct_seed):
    O, eng = oracle, engines[curve]
    F = GD.Field(O, curve)
    m = 3
    proof, V, pubs = product_prove(E, eng, F, struct_seed, 5, m, two_phase, gpu_commit=(struct_seed == 12))
    ref, Vo, pubs_o = oracle_prove(O, curve, F, struct_seed, 5, m, two_phase)
    assert (V == Vo).all() and pubs == pubs_o
    assert proof == ref, "proof bytes differ from the oracle's Prover on the same gadget"
    # Verifier::verify through the generic recorder; the oracle's Verifier agrees
    assert product_verifier(E, curve, F, struct_seed, V, pubs, two_phase).verify(eng, proof) == 0
    assert oracle_verifier(O, curve, F, struct_seed, V, pubs, two_phase).verify(GENS, proof) == 0
    # a wrong public constant, a wrong commitment, a flipped proof byte: VerificationError on both sides
    bad_pubs = list(pubs)
    bad_pubs[0] = (bad_pubs[0] + 1) % F.p
    assert product_verifier(E, curve, F, struct_seed, V, bad_pubs, two_phase).verify(eng, proof) == -4
    assert oracle_verifier(O, curve, F, struct_seed, V, bad_pubs, two_phase).verify(GENS, proof) == O.E_VERIFICATION
    V2 = V.copy()
    V2[[0, 1]] = V2[[1, 0]]
    assert product_verifier(E, curve, F, struct_seed, V2, pubs, two_phase).verify(eng, proof) == -4
    bad = bytearray(proof)
    bad[11 * 33 + 3] ^= 1      # t_x
    assert product_verifier(E, curve, F, struct_seed, V, pubs, two_phase).verify(eng, bytes(bad)) == -4


def test_prover_reports_missing_assignment(E, engines, oracle):
    """R1CSError::MissingAssignment (prover.rs:142,168): allocate(None) on a prover"""
    t = E.HostTranscript(LABEL)
    p = E.ProverCS(0, t)
    with pytest.raises(E.ArkbpError) as ei:
        p.allocate(None)
    assert ei.value.code == -7
    with pytest.raises(E.ArkbpError) as ei:
        p.allocate_multiplier(None)
    assert ei.value.code == -7
    with pytest.raises(E.ArkbpError):     # challenge_scalar exists only inside the randomized phase
        p.challenge_scalar(b"x")


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("two_phase", [False, True])
def test_generic_batch_verify_matches_oracle_point_for_point(E, engines, oracle, curve, two_phase):
    """batch_verify over instances of ONE gadget with different witnesses / public constants / challenges, some of them recorded
    per instance and some as like-instances; the mega-check value must equal the oracle's MSM — identity for a valid batch, the
    same non-identity point when one instance is wrong."""
    O, eng = oracle, engines[curve]
    F = GD.Field(O, curve)
    struct_seed, m = 21, 2
    proofs, Vs, pubs = [], [], []
    for w in range(5):
        pr, V, pb = product_prove(E, eng, F, struct_seed, 40 + w, m, two_phase)
        proofs.append(pr); Vs.append(V); pubs.append(pb)
    # a second gadget in the same batch (another structure, another size)
    pr2, V2, pb2 = product_prove(E, eng, F, 22, 77, 1, two_phase, n_mul=3, n_extra=2)
    alphas = O.fe_rand(O.fid(curve, True), bytes([9]) * 32, 7)

    def run(which_bad):
        pv, ov, pf = [], [], []
        for w in range(5):
            pb = list(pubs[w])
            if w == which_bad:
                pb[-1] = (pb[-1] + 5) % F.p
            pv.append(product_verifier(E, curve, F, struct_seed, Vs[w], pb, two_phase))
            ov.append(oracle_verifier(O, curve, F, struct_seed, Vs[w], pb, two_phase))
            pf.append(proofs[w])
        # instance 0 once more as a like-instance of pv[0] (shares its recording: same publics, own transcript + commitments)
        pv.append(product_verifier(E, curve, F, struct_seed, Vs[0], pubs[0], two_phase, like=pv[0]))
        ov.append(oracle_verifier(O, curve, F, struct_seed, Vs[0], list(pubs[0]) if which_bad != 0 else [*pubs[0][:-1], (pubs[0][-1] + 5) % F.p], two_phase))
        pf.append(proofs[0])
        pv.append(product_verifier(E, curve, F, 22, V2, pb2, two_phase, n_mul=3, n_extra=2))
        ov.append(oracle_verifier(O, curve, F, 22, V2, pb2, two_phase, n_mul=3, n_extra=2))
        pf.append(pr2)
        rc, pt = E.batch_verify_cs(eng, pv, pf, alphas, want_point=True)
        orc, opt = O.batch_verify_cs(curve, ov, pf, GENS, alphas)
        return rc, pt, orc, opt

    rc, pt, orc, opt = run(None)
    assert rc == 0 and orc == 0 and not pt.any() and not opt.any()
    rc, pt, orc, opt = run(3)
    assert rc == -4 and orc == O.E_VERIFICATION
    assert pt.any() and (pt == opt).all(), "the failing batch's mega-check point differs from the oracle's MSM"
    rc, pt, orc, opt = run(0)    # the bad instance is also the like-source: the like-instance shares the bad constant
    assert rc == -4 and orc == O.E_VERIFICATION and (pt == opt).all()


@pytest.mark.parametrize("two_phase", [False, True])
def test_many_like_instances_one_template(E, engines, oracle, two_phase):
    """600 instances of one gadget: 1 recorded + 599 like-instances with their own commitments and proofs (3 distinct proofs
    round-robin), across two 512-blocks; then one tampered -> VerificationError"""
    curve = 0
    O, eng = oracle, engines[curve]
    F = GD.Field(O, curve)
    kw = dict(n_mul=20, n_extra=0, n_alloc=1)     # no public constants: like-instances share everything but V and the proof
    items = [product_prove(E, eng, F, 31, 60 + w, 2, two_phase, **kw) for w in range(3)]
    alphas = O.fe_rand(O.fid(curve, True), bytes([4]) * 32, 600)

    def build(tamper=None):
        vs, pf = [], []
        v0 = product_verifier(E, curve, F, 31, items[0][1], [], two_phase, **kw)
        vs.append(v0); pf.append(items[0][0])
        for k in range(1, 600):
            pr, V, _ = items[k % 3]
            vs.append(product_verifier(E, curve, F, 31, V, [], two_phase, like=v0))
            pf.append(pr)
        if tamper is not None:
            b = bytearray(pf[tamper]); b[-40] ^= 2; pf[tamper] = bytes(b)
        return vs, pf

    vs, pf = build()
    assert E.batch_verify_cs(eng, vs, pf, alphas) == 0
    vs, pf = build(tamper=555)
    assert E.batch_verify_cs(eng, vs, pf, alphas) == -4


def test_transcript_state_roundtrip_and_borrowing(E, engines, oracle):
    """the recorder BORROWS the caller's transcript (`T: BorrowMut<Transcript>`): after prove() the caller's handle has absorbed
    the whole proof, like prove_and_return_transcript; the 203-byte STROBE state moves across unchanged"""
    O, eng = oracle, engines[0]
    F = GD.Field(O, 0)
    t = E.HostTranscript(LABEL)
    before = E.transcript_state(t)
    p = E.ProverCS(0, t)
    V, vars_ = p.commit([F.w(5)], [F.w(7)])
    wit = GD.Witness(F); wit.val[vars_[0]] = 5
    GD.random_program(p, F, 3, wit, vars_, n_mul=2, n_extra=0, n_alloc=0)
    p.prove(eng, bytes(32))
    after = E.transcript_state(t)
    assert before != after and len(after) == 203
    t2 = E.transcript_from_state(after)
    assert t2.challenge_bytes(b"c", 32) == t.challenge_bytes(b"c", 32)
    # the oracle's transcript after the same proof is in the same state: same next challenge
    po = O.ProverCS(0, LABEL).start()
    Vo, vo = po.commit([F.w(5)], [F.w(7)])
    wo = GD.Witness(F); wo.val[vo[0]] = 5
    GD.random_program(po, F, 3, wo, vo, n_mul=2, n_extra=0, n_alloc=0)
    po.prove(GENS, bytes(32))
    t3 = E.transcript_from_state(after)
    assert t3.challenge_bytes(b"c", 32) == po.transcript().challenge_bytes(b"c", 32)


def test_batch_verify_argument_checks(E, engines, oracle):
    """(a) equal weights are refused for more than one instance (verifier.rs:649 draws one per instance), (b) the same
    verifier handle twice is refused at ANY batch size, (c) a like-instance that replayed fewer commitments than the shared
    constraints name is an error, not a silently weaker statement."""
    curve = 0
    O, eng = oracle, engines[curve]
    F = GD.Field(O, curve)
    kw = dict(n_mul=6, n_extra=0, n_alloc=1)
    pr, V, _ = product_prove(E, eng, F, 41, 70, 2, False, **kw)
    lib, C = E.lib(), E.C

    def raw_batch(vs, proofs, alphas):
        hs = (C.c_void_p * len(vs))(*[v.h for v in vs])
        lens = (C.c_size_t * len(vs))(*[len(p) for p in proofs])
        al = None if alphas is None else E.ptr(E.u64arr(alphas, 4))
        return lib.bp_r1cs_batch_verify(eng.ctx, C.c_size_t(len(vs)), hs, b"".join(proofs), lens, al, None, None)

    # (a) NULL alphas: fine for one instance (Verifier::verify), BP_E_ARG for two; the Python wrapper refuses as well
    v0 = product_verifier(E, curve, F, 41, V, [], False, **kw)
    assert raw_batch([v0], [pr], None) == 0
    va, vb = product_verifier(E, curve, F, 41, V, [], False, **kw), product_verifier(E, curve, F, 41, V, [], False, **kw)
    assert raw_batch([va, vb], [pr, pr], None) == -1
    with pytest.raises(ValueError):
        E.batch_verify_cs(eng, [va, vb], [pr, pr], None)
    alphas = O.fe_rand(O.fid(curve, True), bytes([3]) * 32, 5000)
    assert raw_batch([va, vb], [pr, pr], alphas[:2]) == 0           # (the refused calls consumed nothing)
    # (b) duplicates in a batch above the old 4096 cut-off
    src = product_verifier(E, curve, F, 41, V, [], False, **kw)
    many = [src] + [product_verifier(E, curve, F, 41, V, [], False, like=src) for _ in range(4199)]
    many[4150] = many[7]
    assert raw_batch(many, [pr] * len(many), alphas[: len(many)]) == -1
    # (c) a like-instance with one commitment missing
    src2 = product_verifier(E, curve, F, 41, V, [], False, **kw)
    t = E.HostTranscript(LABEL)
    t.append_message(b"dom-sep", b"generic gadget v1")
    short = E.VerifierCS(curve, t, like=src2)
    short.commit(V[:1])
    assert raw_batch([src2, short], [pr, pr], alphas[:2]) == -1
    assert short.verify(eng, pr) != 0
