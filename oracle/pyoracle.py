"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes loader for oracle/liboracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity status: "parity unpinned" (see oracle/field.hpp).

Conventions: field elements are numpy uint64[4] (Montgomery form, little-endian limbs), points are
uint64[8] (x || y, identity = all zero); vectors are C-contiguous arrays of those.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SECQ, ZORRO = 0, 1
SC_SHUFFLE, SC_RANGE, SC_EXAMPLE, SC_SQUARE_CHAIN, SC_MULTI_RANGE = 0, 1, 2, 3, 4
OK, E_VERIFICATION, E_GENS_LENGTH, E_MISSING_ASSIGNMENT, E_FORMAT, E_GADGET = 0, 1, 2, 3, 4, 5   # protocol.hpp `enum Err`


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_msm_timed.restype = C.c_double
        _LIB.orc_transcript_new.restype = C.c_void_p
        _LIB.orc_transcript_clone.restype = C.c_void_p
        _LIB.orc_init()
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u64(n):
    return np.zeros(n, dtype=np.uint64)


def set_msm_threads(n):
    """threads over the Pippenger windows of every msm call (the reference's optional `parallel` feature); 1 = its default"""
    lib().orc_set_msm_threads(int(n))


def fid(curve, scalar_field):
    return 2 * curve + (1 if scalar_field else 0)


# ---- fields -------------------------------------------------------------------------------
def modulus(f):
    o = _u64(4)
    lib().orc_fe_modulus(f, _p(o))
    return limbs_to_int(o)


def limbs_to_int(a):
    return sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(a, dtype=np.uint64).reshape(-1)[:4]))


def int_to_limbs(x):
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def fe_from_int(f, x):
    o = _u64(4)
    lib().orc_fe_from_canon(f, _p(int_to_limbs(x % modulus(f))), _p(o))
    return o


def fe_to_int(f, a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    o = _u64(4)
    lib().orc_fe_to_canon(f, _p(a), _p(o))
    return limbs_to_int(o)


def fe_op(name, f, a, b=None):
    o = _u64(4)
    a = np.ascontiguousarray(a, dtype=np.uint64)
    fn = getattr(lib(), "orc_fe_" + name)
    if b is None:
        rc = fn(f, _p(a), _p(o))
        return o if rc == 0 else None
    b = np.ascontiguousarray(b, dtype=np.uint64)
    fn(f, _p(a), _p(b), _p(o))
    return o


def fe_rand(f, seed, count):
    o = _u64(4 * count)
    lib().orc_fe_rand(f, bytes(seed), C.c_size_t(count), _p(o))
    return o.reshape(count, 4)


# ---- bytes --------------------------------------------------------------------------------
def sha3_512(msg):
    out = C.create_string_buffer(64)
    lib().orc_sha3_512(bytes(msg), C.c_size_t(len(msg)), out)
    return out.raw


def chacha20_words(seed, n):
    o = np.zeros(n, dtype=np.uint32)
    lib().orc_chacha20_words(bytes(seed), C.c_size_t(n), _p(o))
    return o


class Transcript:
    def __init__(self, label=None, handle=None):
        self.h = C.c_void_p(handle if handle is not None else lib().orc_transcript_new(bytes(label), C.c_size_t(len(label))))

    def clone(self):
        return Transcript(handle=lib().orc_transcript_clone(self.h))

    def __del__(self):
        try:
            if not getattr(self, "_borrowed", False):
                lib().orc_transcript_free(self.h)
        except Exception:
            pass

    def append_message(self, label, msg):
        lib().orc_transcript_append_message(self.h, bytes(label), bytes(msg), C.c_size_t(len(msg)))

    def append_u64(self, label, x):
        lib().orc_transcript_append_u64(self.h, bytes(label), C.c_uint64(x))

    def challenge_bytes(self, label, n):
        out = C.create_string_buffer(n)
        lib().orc_transcript_challenge_bytes(self.h, bytes(label), out, C.c_size_t(n))
        return out.raw

    def append_point(self, curve, label, xy):
        lib().orc_transcript_append_point(curve, self.h, bytes(label), _p(np.ascontiguousarray(xy, dtype=np.uint64)))

    def append_scalar(self, curve, label, s):
        lib().orc_transcript_append_scalar(curve, self.h, bytes(label), _p(np.ascontiguousarray(s, dtype=np.uint64)))

    def challenge_scalar(self, curve, label):
        o = _u64(4)
        lib().orc_transcript_challenge_scalar(curve, self.h, bytes(label), _p(o))
        return o

    def rng_draws(self, curve, witness, seed, count):
        w = np.ascontiguousarray(witness, dtype=np.uint64).reshape(-1, 4)
        o = _u64(4 * count)
        lib().orc_transcript_rng_draws(curve, self.h, _p(w), C.c_size_t(len(w)), bytes(seed), C.c_size_t(count), _p(o))
        return o.reshape(count, 4)


# ---- group --------------------------------------------------------------------------------
def generator(curve):
    o = _u64(8)
    lib().orc_generator(curve, _p(o))
    return o


def on_curve(curve, xy):
    return bool(lib().orc_on_curve(curve, _p(np.ascontiguousarray(xy, dtype=np.uint64))))


def point_add(curve, p, q):
    o = _u64(8)
    lib().orc_point_add(curve, _p(np.ascontiguousarray(p, dtype=np.uint64)), _p(np.ascontiguousarray(q, dtype=np.uint64)), _p(o))
    return o


def scalar_mul(curve, p, s):
    o = _u64(8)
    lib().orc_scalar_mul(curve, _p(np.ascontiguousarray(p, dtype=np.uint64)), _p(np.ascontiguousarray(s, dtype=np.uint64)), _p(o))
    return o


def point_ser(curve, p, compressed):
    out = C.create_string_buffer(33 if compressed else 65)
    lib().orc_point_ser(curve, _p(np.ascontiguousarray(p, dtype=np.uint64)), int(compressed), out)
    return out.raw


def point_deser_compressed(curve, b):
    o = _u64(8)
    rc = lib().orc_point_deser_compressed(curve, bytes(b), _p(o))
    return o if rc == 0 else None


def msm(curve, bases, scalars, timed=False):
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    assert len(bases) == len(scalars)
    o = _u64(8)
    if timed:
        t = lib().orc_msm_timed(curve, _p(bases), _p(scalars), C.c_size_t(len(bases)), _p(o))
        return o, t
    lib().orc_msm(curve, _p(bases), _p(scalars), C.c_size_t(len(bases)), _p(o))
    return o


def pedersen_default(curve):
    b, bb = _u64(8), _u64(8)
    lib().orc_pedersen_default(curve, _p(b), _p(bb))
    return b, bb


def pedersen_commit(curve, v, blind):
    o = _u64(8)
    lib().orc_pedersen_commit(curve, _p(np.ascontiguousarray(v, dtype=np.uint64)), _p(np.ascontiguousarray(blind, dtype=np.uint64)), _p(o))
    return o


def bp_gens(curve, cap):
    g, h = _u64(8 * cap), _u64(8 * cap)
    lib().orc_bp_gens(curve, C.c_size_t(cap), _p(g), _p(h))
    return g.reshape(cap, 8), h.reshape(cap, 8)


def bp_gens_party(curve, cap, party):
    g, h = _u64(8 * cap), _u64(8 * cap)
    lib().orc_bp_gens_party(curve, C.c_size_t(cap), C.c_size_t(party), _p(g), _p(h))
    return g.reshape(cap, 8), h.reshape(cap, 8)


# ---- IPA ---------------------------------------------------------------------------------
def ipa_create(curve, tr, Q, Gf, Hf, G, H, a, b):
    n = len(G)
    lg = max(n.bit_length() - 1, 0)
    L, R, ao, bo = _u64(8 * max(lg, 1)), _u64(8 * max(lg, 1)), _u64(4), _u64(4)
    arrs = [np.ascontiguousarray(x, dtype=np.uint64) for x in (Q, Gf, Hf, G, H, a, b)]
    k = lib().orc_ipa_create(curve, tr.h, *[_p(x) for x in arrs], C.c_size_t(n), _p(L), _p(R), _p(ao), _p(bo))
    return L.reshape(-1, 8)[:k], R.reshape(-1, 8)[:k], ao, bo


def ipa_verify(curve, tr, n, Gf, Hf, P, Q, G, H, L, R, a, b):
    arrs = [np.ascontiguousarray(x, dtype=np.uint64) for x in (Gf, Hf, P, Q, G, H, L, R)]
    lgn = len(np.asarray(L).reshape(-1, 8)) if np.asarray(L).size else 0
    return lib().orc_ipa_verify(curve, tr.h, C.c_size_t(n), *[_p(x) for x in arrs], C.c_size_t(lgn),
                                _p(np.ascontiguousarray(a, dtype=np.uint64)), _p(np.ascontiguousarray(b, dtype=np.uint64)))


def ipa_verification_scalars(curve, tr, n, L, R):
    L = np.ascontiguousarray(L, dtype=np.uint64).reshape(-1, 8)
    R = np.ascontiguousarray(R, dtype=np.uint64).reshape(-1, 8)
    lgn = len(L)
    us, uis, s = _u64(4 * max(lgn, 1)), _u64(4 * max(lgn, 1)), _u64(4 * n)
    rc = lib().orc_ipa_verification_scalars(curve, tr.h, C.c_size_t(n), _p(L), _p(R), C.c_size_t(lgn), _p(us), _p(uis), _p(s))
    if rc:
        return None
    return us.reshape(-1, 4)[:lgn], uis.reshape(-1, 4)[:lgn], s.reshape(n, 4)


# ---- R1CS scenarios ------------------------------------------------------------------------
def _params(params):
    a = np.zeros(8, dtype=np.uint64)
    a[: len(params)] = np.array(params, dtype=np.uint64)
    return a


class Proved:
    def __init__(self, rc, proof=b"", commitments=None, publics=None, t_prove=0.0, t_setup=0.0):
        self.rc, self.proof, self.commitments, self.publics = rc, proof, commitments, publics
        self.t_prove, self.t_setup = t_prove, t_setup


def r1cs_prove(curve, scenario, params, seed, gens_cap, m_cap=None):
    prm = _params(params)
    if m_cap is None:
        m_cap = 2 * int(prm[0]) + 8
    buf = C.create_string_buffer(1 << 16)
    plen = C.c_size_t(len(buf))
    commits = _u64(8 * m_cap)
    m = C.c_size_t(0)
    pubs = _u64(4 * 8)
    npub = C.c_size_t(0)
    timing = (C.c_double * 2)()
    rc = lib().orc_r1cs_prove(curve, scenario, _p(prm), bytes(seed), C.c_size_t(gens_cap), buf, C.byref(plen), _p(commits), C.c_size_t(m_cap),
                              C.byref(m), _p(pubs), C.byref(npub), timing)
    if rc:
        return Proved(rc)
    return Proved(0, buf.raw[: plen.value], commits.reshape(-1, 8)[: m.value].copy(), pubs.reshape(-1, 4)[: npub.value].copy(), timing[0], timing[1])


def r1cs_verify(curve, scenario, params, gens_cap, proof, commitments, publics, timing=None):
    prm = _params(params)
    cm = np.ascontiguousarray(commitments, dtype=np.uint64).reshape(-1, 8)
    pb = np.ascontiguousarray(publics, dtype=np.uint64).reshape(-1, 4)
    t = (C.c_double * 1)()
    rc = lib().orc_r1cs_verify(curve, scenario, _p(prm), C.c_size_t(gens_cap), bytes(proof), C.c_size_t(len(proof)), _p(cm), C.c_size_t(len(cm)),
                               _p(pb), C.c_size_t(len(pb)), t)
    if timing is not None:
        timing.append(t[0])
    return rc


def r1cs_verification_scalars(curve, scenario, params, gens_cap, proof, commitments, publics, cap):
    prm = _params(params)
    cm = np.ascontiguousarray(commitments, dtype=np.uint64).reshape(-1, 8)
    pb = np.ascontiguousarray(publics, dtype=np.uint64).reshape(-1, 4)
    out = _u64(4 * cap)
    cnt = C.c_size_t(0)
    rc = lib().orc_r1cs_verification_scalars(curve, scenario, _p(prm), C.c_size_t(gens_cap), bytes(proof), C.c_size_t(len(proof)), _p(cm),
                                             C.c_size_t(len(cm)), _p(pb), C.c_size_t(len(pb)), _p(out), C.c_size_t(cap), C.byref(cnt))
    if rc:
        return rc, None
    return 0, out.reshape(-1, 4)[: cnt.value].copy()


def batch_verify(curve, instances, gens_cap, alpha_seed, timing=None):
    """instances: list of (scenario, params, proof_bytes, commitments, publics)"""
    n = len(instances)
    scen = (C.c_int * n)(*[i[0] for i in instances])
    prm = np.concatenate([_params(i[1]) for i in instances])
    proofs = b"".join(i[2] for i in instances)
    plens = (C.c_size_t * n)(*[len(i[2]) for i in instances])
    cms = np.ascontiguousarray(np.concatenate([np.asarray(i[3], dtype=np.uint64).reshape(-1, 8) for i in instances]))
    ms = (C.c_size_t * n)(*[len(np.asarray(i[3]).reshape(-1, 8)) for i in instances])
    pubs_l = [np.asarray(i[4], dtype=np.uint64).reshape(-1, 4) for i in instances]
    pubs = np.ascontiguousarray(np.concatenate(pubs_l + [np.zeros((1, 4), dtype=np.uint64)]))
    npubs = (C.c_size_t * n)(*[len(p) for p in pubs_l])
    t = (C.c_double * 1)()
    rc = lib().orc_batch_verify(curve, C.c_size_t(n), scen, _p(prm), C.c_size_t(gens_cap), proofs, plens, _p(cms), ms, _p(pubs), npubs,
                                bytes(alpha_seed), t)
    if timing is not None:
        timing.append(t[0])
    return rc


# ---- the reference's ConstraintSystem trait on the restated Prover / Verifier (capi.cpp orc_cs_*) ---------------------------
# Same shapes as ark_bulletproofs_amd.engine.ProverCS / VerifierCS, so one gadget function drives both sides of a parity test.
VAR_COMMITTED, VAR_MULT_LEFT, VAR_MULT_RIGHT, VAR_MULT_OUT, VAR_ONE = 0, 1, 2, 3, 4
ONE_VAR = (VAR_ONE, 0)
_RANDOMIZE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


def _lc_arrays(lc):
    n = len(lc)
    vars_ = np.zeros((max(n, 1), 2), dtype=np.uint32)
    coefs = np.zeros((max(n, 1), 4), dtype=np.uint64)
    for i, (v, c) in enumerate(lc):
        vars_[i] = v
        coefs[i] = np.asarray(c, dtype=np.uint64).reshape(4)
    return vars_, coefs, n


def _vars_out(arr):
    return [(int(arr[i, 0]), int(arr[i, 1])) for i in range(len(arr))]


class _OracleCS:
    def __init__(self, curve, label, proving):
        lib().orc_cs_new.restype = C.c_void_p
        lib().orc_cs_transcript.restype = C.c_void_p
        self.curve = curve
        self._pre = Transcript(label)            # the caller's transcript before Prover::new / Verifier::new
        self.h = None
        self.proving = proving
        self._cbs, self._cb_errors = [], []

    def transcript(self):
        """before start(): the transcript the recorder will be created on; after: the recorder's own"""
        if self.h is None:
            return self._pre
        t = Transcript.__new__(Transcript)
        t.h = C.c_void_p(lib().orc_cs_transcript(self.h))
        t._borrowed = True
        return t

    def start(self):
        self.h = C.c_void_p(lib().orc_cs_new(self.curve, int(self.proving), self._pre.h, b"", C.c_size_t(0)))
        return self

    def __del__(self):
        try:
            if self.h:
                lib().orc_cs_free(self.h)
        except Exception:
            pass

    def multiply(self, left, right):
        lv, lc, nl = _lc_arrays(left)
        rv, rc, nr = _lc_arrays(right)
        out = np.zeros((3, 2), dtype=np.uint32)
        assert lib().orc_cs_multiply(self.h, _p(lv), _p(lc), C.c_size_t(nl), _p(rv), _p(rc), C.c_size_t(nr), _p(out)) == 0
        return _vars_out(out)

    def allocate(self, assignment=None):
        out = np.zeros((1, 2), dtype=np.uint32)
        a = None if assignment is None else np.ascontiguousarray(assignment, dtype=np.uint64).reshape(4)
        rc = lib().orc_cs_allocate(self.h, _p(a) if a is not None else None, _p(out))
        if rc:
            raise OracleError(rc)
        return _vars_out(out)[0]

    def allocate_multiplier(self, assignments=None):
        out = np.zeros((3, 2), dtype=np.uint32)
        if assignments is None:
            rc = lib().orc_cs_allocate_multiplier(self.h, None, None, _p(out))
        else:
            l, r = [np.ascontiguousarray(x, dtype=np.uint64).reshape(4) for x in assignments]
            rc = lib().orc_cs_allocate_multiplier(self.h, _p(l), _p(r), _p(out))
        if rc:
            raise OracleError(rc)
        return _vars_out(out)

    def constrain(self, lc):
        v, c, n = _lc_arrays(lc)
        assert lib().orc_cs_constrain(self.h, _p(v), _p(c), C.c_size_t(n)) == 0

    def specify_randomized_constraints(self, fn):
        me = self

        def thunk(_user, _handle):
            try:
                fn(me)
                return 0
            except Exception as e:
                me._cb_errors.append(e)
                return -100

        cb = _RANDOMIZE_CB(thunk)
        self._cbs.append(cb)
        assert lib().orc_cs_specify_randomized_constraints(self.h, cb, None) == 0

    def challenge_scalar(self, label):
        out = _u64(4)
        lib().orc_cs_challenge_scalar(self.h, bytes(label) + b"\0", _p(out))
        return out


class OracleError(RuntimeError):
    def __init__(self, code):
        self.code = code
        super().__init__("oracle error %d" % code)


class ProverCS(_OracleCS):
    def __init__(self, curve, label):
        super().__init__(curve, label, True)

    def commit(self, v, v_blinding):
        v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
        b = np.ascontiguousarray(v_blinding, dtype=np.uint64).reshape(-1, 4)
        V = np.zeros((len(v), 8), dtype=np.uint64)
        vars_ = np.zeros((len(v), 2), dtype=np.uint32)
        for i in range(len(v)):
            assert lib().orc_prover_commit(self.h, _p(v[i]), _p(b[i]), _p(V[i]), _p(vars_[i])) == 0
        return V, _vars_out(vars_)

    def prove(self, gens_cap, rng_bytes):
        buf = C.create_string_buffer(1 << 16)
        plen = C.c_size_t(len(buf))
        rc = lib().orc_prover_prove(self.h, bytes(rng_bytes), C.c_size_t(gens_cap), buf, C.byref(plen))
        if self._cb_errors:
            raise self._cb_errors[0]
        if rc:
            raise OracleError(rc)
        return buf.raw[: plen.value]


class VerifierCS(_OracleCS):
    def __init__(self, curve, label):
        super().__init__(curve, label, False)

    def commit(self, V):
        V = np.ascontiguousarray(V, dtype=np.uint64).reshape(-1, 8)
        vars_ = np.zeros((len(V), 2), dtype=np.uint32)
        for i in range(len(V)):
            assert lib().orc_verifier_commit(self.h, _p(V[i]), _p(vars_[i])) == 0
        return _vars_out(vars_)

    def verify(self, gens_cap, proof):
        rc = lib().orc_verifier_verify(self.h, C.c_size_t(gens_cap), bytes(proof), C.c_size_t(len(proof)))
        if self._cb_errors:
            raise self._cb_errors[0]
        return rc


def batch_verify_cs(curve, verifiers, proofs, gens_cap, alphas):
    """returns (status, mega-check point)"""
    n = len(verifiers)
    hs = (C.c_void_p * max(n, 1))(*[v.h for v in verifiers])
    blob = b"".join(bytes(p) for p in proofs)
    lens = (C.c_size_t * max(n, 1))(*[len(p) for p in proofs])
    al = np.ascontiguousarray(alphas, dtype=np.uint64).reshape(-1, 4)
    pt = _u64(8)
    rc = lib().orc_cs_batch_verify(curve, C.c_size_t(n), hs, C.c_size_t(gens_cap), blob, lens, _p(al), _p(pt))
    for v in verifiers:
        if v._cb_errors:
            raise v._cb_errors[0]
    return rc, pt


def exp_iter(f, x, n):
    out = np.zeros((n, 4), dtype=np.uint64)
    lib().orc_exp_iter(f, _p(np.ascontiguousarray(x, dtype=np.uint64)), C.c_size_t(n), _p(out))
    return out


def inner_product(f, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    out = _u64(4)
    lib().orc_inner_product(f, _p(a), _p(b), C.c_size_t(len(a)), _p(out))
    return out


def batch_verify_point(curve, instances, gens_cap, alpha_seed):
    """batch_verify over scenario instances [(scenario, params, proof, commitments, publics)]: (status, mega-check point)"""
    n = len(instances)
    scen = (C.c_int * n)(*[i[0] for i in instances])
    prm = np.zeros((n, 8), dtype=np.uint64)
    for k, i in enumerate(instances):
        prm[k, : len(i[1])] = np.array(i[1], dtype=np.uint64)
    proofs = b"".join(i[2] for i in instances)
    plens = (C.c_size_t * n)(*[len(i[2]) for i in instances])
    cm_l = [np.asarray(i[3], dtype=np.uint64).reshape(-1, 8) for i in instances]
    cms = np.ascontiguousarray(np.concatenate(cm_l))
    ms = (C.c_size_t * n)(*[len(c) for c in cm_l])
    pubs_l = [np.asarray(i[4], dtype=np.uint64).reshape(-1, 4) for i in instances]
    pubs = np.ascontiguousarray(np.concatenate(pubs_l + [np.zeros((1, 4), dtype=np.uint64)]))
    npubs = (C.c_size_t * n)(*[len(p) for p in pubs_l])
    pt = _u64(8)
    rc = lib().orc_batch_verify_point(curve, C.c_size_t(n), scen, _p(prm), C.c_size_t(gens_cap), proofs, plens, _p(cms), ms, _p(pubs), npubs, bytes(alpha_seed), _p(pt))
    return rc, pt
