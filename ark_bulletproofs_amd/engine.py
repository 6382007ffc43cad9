"""Engine: one bp_ctx (one GPU, one HIP stream) and typed wrappers over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ArkbpError, check, lib, ptr, u64arr  # noqa: F401

SECQ256K1, ZORRO = 0, 1


class DeviceBuffer:
    """HBM allocation owned by an Engine."""

    def __init__(self, eng, nbytes):
        self.eng, self.nbytes = eng, nbytes
        p = C.c_void_p()
        check(lib().bp_dev_alloc(eng.ctx, C.c_size_t(nbytes), C.byref(p)), "bp_dev_alloc")
        self.ptr = p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(lib().bp_dev_upload(self.eng.ctx, self.ptr, ptr(arr), C.c_size_t(arr.nbytes)), "bp_dev_upload")
        return self

    def download(self, dtype=np.uint64, nbytes=None):
        nbytes = self.nbytes if nbytes is None else nbytes
        out = np.zeros(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        check(lib().bp_dev_download(self.eng.ctx, ptr(out), self.ptr, C.c_size_t(nbytes)), "bp_dev_download")
        return out

    def free(self):
        if self.ptr:
            lib().bp_dev_free(self.eng.ctx, self.ptr)
            self.ptr = None


class Engine:
    def __init__(self, curve=SECQ256K1, device=0):
        self.curve = curve
        self.ctx = C.c_void_p()
        check(lib().bp_ctx_create(curve, device, C.byref(self.ctx)), "bp_ctx_create")

    @classmethod
    def host_only(cls, curve=SECQ256K1, gens_capacity=256):
        """a ctx WITHOUT a device (bp_debug_ctx_create_hostonly): the host side of batch verification as a dry run, for sanitizer
        builds on machines with no GPU.  Nothing is verified in that mode."""
        self = cls.__new__(cls)
        self.curve = curve
        self.ctx = C.c_void_p()
        check(lib().bp_debug_ctx_create_hostonly(curve, C.c_size_t(gens_capacity), C.byref(self.ctx)), "bp_debug_ctx_create_hostonly")
        return self

    def close(self):
        if self.ctx:
            lib().bp_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- memory -------------------------------------------------------------------------------
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def upload_points(self, pts_xy):
        """ark-layout affine points (n x 8 u64) -> resident engine layout"""
        pts = u64arr(pts_xy, 8)
        buf = DeviceBuffer(self, max(pts.nbytes, 64)).upload(pts)
        check(lib().bp_points_import(self.ctx, buf.ptr, buf.ptr, C.c_size_t(len(pts))), "bp_points_import")
        return buf

    def upload_scalars(self, scalars):
        sc = u64arr(scalars, 4)
        return DeviceBuffer(self, max(sc.nbytes, 32)).upload(sc)

    def sync(self):
        check(lib().bp_ctx_sync(self.ctx), "bp_ctx_sync")

    # ---- VariableBaseMSM::msm -------------------------------------------------------------------
    def msm(self, bases_xy, scalars, canonical=False):
        b, s = u64arr(bases_xy, 8), u64arr(scalars, 4)
        if len(b) != len(s):
            raise ValueError("msm: %d bases vs %d scalars" % (len(b), len(s)))  # ark: Err(min_len)
        out = np.zeros(8, dtype=np.uint64)
        check(lib().bp_msm(self.ctx, ptr(b), ptr(s), C.c_size_t(len(b)), int(canonical), ptr(out)), "bp_msm")
        return out

    def msm_gens(self, n, scalars, use_G=True, use_H=True, off=0, extra_bases=None, canonical=False):
        """MSM over the resident generator tables: G[off:off+n] || H[off:off+n] || extra_bases, scalars in that order"""
        ex = u64arr(extra_bases, 8) if extra_bases is not None and len(extra_bases) else np.zeros((0, 8), dtype=np.uint64)
        s = u64arr(scalars, 4)
        if len(s) != (n if use_G else 0) + (n if use_H else 0) + len(ex):
            raise ValueError("msm_gens: bases and scalars differ in length")
        out = np.zeros(8, dtype=np.uint64)
        check(lib().bp_msm_gens(self.ctx, int(use_G), int(use_H), C.c_size_t(off), C.c_size_t(n), ptr(ex) if len(ex) else None, C.c_size_t(len(ex)),
                                ptr(s) if len(s) else None, int(canonical), ptr(out)), "bp_msm_gens")
        return out

    def msm_dev(self, d_bases, d_scalars, n, canonical=False):
        out = np.zeros(8, dtype=np.uint64)
        check(lib().bp_msm_dev(self.ctx, d_bases.ptr, d_scalars.ptr, C.c_size_t(n), int(canonical), ptr(out)), "bp_msm_dev")
        return out

    def set_window_shard(self, rank, world, reduce_fn=None):
        """window-sharded mode (bp_ctx_set_window_shard): reduce_fn(xy: (8,) u64) -> (8,) u64 = sum of all ranks' partial points"""
        if world <= 1:
            check(lib().bp_ctx_set_window_shard(self.ctx, 0, 1, None, None), "bp_ctx_set_window_shard")
            self._shard_cb = None
            return
        errs = self._shard_errors = []

        def cb(_user, xy):
            try:
                out = reduce_fn(np.array(xy[:8], dtype=np.uint64))
                for i in range(8):
                    xy[i] = int(out[i])
                return 0
            except Exception as e:  # never unwind through C
                errs.append(e)
                return 1

        self._shard_cb = _POINT_REDUCE_CB(cb)   # keep the thunk alive as long as the mode is on
        check(lib().bp_ctx_set_window_shard(self.ctx, int(rank), int(world), self._shard_cb, None), "bp_ctx_set_window_shard")

    def set_shard_allgather(self, gather_fn=None):
        """second collective of the sharded mode (bp_ctx_set_shard_allgather): gather_fn(block: uint8 array) -> (world, len) uint8
        array in rank order.  With it the prover partitions the inner-product argument index-cyclically across the ranks."""
        if gather_fn is None:
            check(lib().bp_ctx_set_shard_allgather(self.ctx, None, None), "bp_ctx_set_shard_allgather")
            self._gather_cb = None
            return
        errs = self._gather_errors = []

        def cb(_user, send, nbytes, recv):
            try:
                blk = np.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=np.uint8)
                out = np.ascontiguousarray(gather_fn(blk.copy()), dtype=np.uint8).reshape(-1)
                C.memmove(recv, out.ctypes.data, out.nbytes)
                return 0
            except Exception as e:  # never unwind through C
                errs.append(e)
                return 1

        self._gather_cb = _ALLGATHER_CB(cb)
        check(lib().bp_ctx_set_shard_allgather(self.ctx, self._gather_cb, None), "bp_ctx_set_shard_allgather")

    def set_tuning(self, knob, value):
        """knob: 0 fold-batch minimum lanes, 1 MSM two-level-sort minimum terms, .. 6 host threads of the ctx's pool (include/arkbp.h BP_TUNE_*)"""
        check(lib().bp_ctx_set_tuning(self.ctx, int(knob), C.c_uint64(int(value))), "bp_ctx_set_tuning")

    # ---- profiling ------------------------------------------------------------------------------
    def set_profiling(self, on=True):
        check(lib().bp_ctx_set_profiling(self.ctx, int(on)), "bp_ctx_set_profiling")

    def reset_profiling(self):
        check(lib().bp_ctx_reset_profiling(self.ctx), "bp_ctx_reset_profiling")

    def kernel_time(self, which):
        ms, cnt = C.c_double(0), C.c_uint64(0)
        check(lib().bp_ctx_kernel_time(self.ctx, which, C.byref(ms), C.byref(cnt)), "bp_ctx_kernel_time")
        return ms.value, cnt.value

    # ---- unit-test hooks ---------------------------------------------------------------------------
    def debug_field_op(self, field, op, a, b):
        a, b = u64arr(a, 4), u64arr(b, 4)
        out = np.zeros_like(a)
        check(lib().bp_debug_field_op(self.ctx, field, op, ptr(a), ptr(b), ptr(out), C.c_size_t(len(a))), "bp_debug_field_op")
        return out

    def debug_point_op(self, op, p, q, k=None):
        p, q = u64arr(p, 8), u64arr(q, 8)
        k = np.zeros((len(p), 4), dtype=np.uint64) if k is None else u64arr(k, 4)
        out = np.zeros_like(p)
        check(lib().bp_debug_point_op(self.ctx, op, ptr(p), ptr(q), ptr(k), ptr(out), C.c_size_t(len(p))), "bp_debug_point_op")
        return out


# ---- InnerProductProof::create ---------------------------------------------------------------------
_POINT_REDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64))
_ALLGATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_CHALLENGE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))


def _ipa_create(self, Q, G_factors, H_factors, G_vec, H_vec, a_vec, b_vec, challenge):
    """challenge(L_xy, R_xy) -> u (4 x u64 Montgomery words): the caller's transcript step
    (append_point L, R; challenge_scalar u — src/inner_product_proof.rs:132-135)."""
    G, H = u64arr(G_vec, 8), u64arr(H_vec, 8)
    n = len(G)
    arrs = [u64arr(Q, 8)] + [u64arr(x, 4) for x in (G_factors, H_factors)] + [G, H] + [u64arr(x, 4) for x in (a_vec, b_vec)]
    for x in arrs[1:]:
        if len(x) != n:
            raise ValueError("ipa_create: vector lengths differ")  # the reference asserts
    lg = max(n.bit_length() - 1, 0)
    L = np.zeros((max(lg, 1), 8), dtype=np.uint64)
    R = np.zeros((max(lg, 1), 8), dtype=np.uint64)
    ao, bo = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    err = []

    def cb(_user, Lp, Rp, up):
        try:
            u = challenge(np.array(Lp[:8], dtype=np.uint64), np.array(Rp[:8], dtype=np.uint64))
            for i in range(4):
                up[i] = int(u[i])
            return 0
        except Exception as e:  # never unwind through C
            err.append(e)
            return 1

    cfn = _CHALLENGE_CB(cb)
    rc = lib().bp_ipa_create(self.ctx, *[ptr(x) for x in arrs], C.c_size_t(n), cfn, None, ptr(L), ptr(R), ptr(ao), ptr(bo))
    if err:
        raise err[0]
    check(rc, "bp_ipa_create")
    return L[:lg], R[:lg], ao, bo


Engine.ipa_create = _ipa_create


# InnerProductProof::create cut at the Fiat-Shamir step (bp_ipa_begin .. bp_ipa_finish): the stepping interface that
# parallel.sharded_ipa_create drives on every rank
def _ipa_begin(self, Q, G_factors, H_factors, G_vec, H_vec, a_vec, b_vec):
    G, H = u64arr(G_vec, 8), u64arr(H_vec, 8)
    n = len(G)
    arrs = [u64arr(Q, 8)] + [u64arr(x, 4) for x in (G_factors, H_factors)] + [G, H] + [u64arr(x, 4) for x in (a_vec, b_vec)]
    for x in arrs[1:]:
        if len(x) != n:
            raise ValueError("ipa_begin: vector lengths differ")
    check(lib().bp_ipa_begin(self.ctx, *[ptr(x) for x in arrs], C.c_size_t(n)), "bp_ipa_begin")


def _ipa_round_LR(self):
    L, R = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    check(lib().bp_ipa_round_LR(self.ctx, ptr(L), ptr(R)), "bp_ipa_round_LR")
    return L, R


def _ipa_round_fold(self, u):
    u = np.ascontiguousarray(u, dtype=np.uint64).reshape(4)
    check(lib().bp_ipa_round_fold(self.ctx, ptr(u)), "bp_ipa_round_fold")


def _ipa_finish(self):
    a, b = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    check(lib().bp_ipa_finish(self.ctx, ptr(a), ptr(b)), "bp_ipa_finish")
    return a, b


def _ipa_export(self, n_max):
    """current (a, b, G, H, gamma_G, gamma_H); G_true = gamma_G * G, H_true = gamma_H * H"""
    a, b = np.zeros((n_max, 4), dtype=np.uint64), np.zeros((n_max, 4), dtype=np.uint64)
    G, H = np.zeros((n_max, 8), dtype=np.uint64), np.zeros((n_max, 8), dtype=np.uint64)
    gG, gH = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    n = C.c_size_t(0)
    check(lib().bp_ipa_export(self.ctx, ptr(a), ptr(b), ptr(G), ptr(H), ptr(gG), ptr(gH), C.byref(n)), "bp_ipa_export")
    return a[: n.value], b[: n.value], G[: n.value], H[: n.value], gG, gH


Engine.ipa_begin = _ipa_begin
Engine.ipa_round_LR = _ipa_round_LR
Engine.ipa_round_fold = _ipa_round_fold
Engine.ipa_finish = _ipa_finish
Engine.ipa_export = _ipa_export


# ---- generators / prover ---------------------------------------------------------------------------
SC_SHUFFLE, SC_RANGE, SC_EXAMPLE, SC_SQUARE_CHAIN, SC_MULTI_RANGE = 0, 1, 2, 3, 4


def _gens_derive(self, cap):
    """BulletproofGens::new(cap, 1) + PedersenGens::default(), installed in HBM"""
    check(lib().bp_gens_derive(self.ctx, C.c_size_t(cap)), "bp_gens_derive")
    self.gens_capacity = cap


def _gens_upload(self, G_xy, H_xy):
    G, H = u64arr(G_xy, 8), u64arr(H_xy, 8)
    assert len(G) == len(H)
    check(lib().bp_gens_upload(self.ctx, ptr(G), ptr(H), C.c_size_t(len(G))), "bp_gens_upload")
    self.gens_capacity = len(G)


def _gens_download(self, n):
    G, H = np.zeros((n, 8), dtype=np.uint64), np.zeros((n, 8), dtype=np.uint64)
    check(lib().bp_gens_download(self.ctx, ptr(G), ptr(H), C.c_size_t(n)), "bp_gens_download")
    return G, H


class Proved:
    def __init__(self, proof, commitments, publics, timing):
        self.proof, self.commitments, self.publics, self.timing = proof, commitments, publics, timing
        self.t_prove, self.t_setup = timing[0], timing[1]


def _prove_scenario(self, scenario, params, seed, m_cap=None):
    prm = np.zeros(8, dtype=np.uint64)
    prm[: len(params)] = np.array(params, dtype=np.uint64)
    if m_cap is None:
        m_cap = 2 * int(prm[0]) + 8
    buf = C.create_string_buffer(1 << 16)
    plen = C.c_size_t(len(buf))
    commits = np.zeros((m_cap, 8), dtype=np.uint64)
    m, npub = C.c_size_t(0), C.c_size_t(0)
    pubs = np.zeros((8, 4), dtype=np.uint64)
    timing = (C.c_double * 8)()
    check(lib().bp_r1cs_prove_scenario(self.ctx, scenario, ptr(prm), bytes(seed), buf, C.byref(plen), ptr(commits), C.c_size_t(m_cap), C.byref(m),
                                       ptr(pubs), C.byref(npub), timing), "bp_r1cs_prove_scenario")
    return Proved(buf.raw[: plen.value], commits[: m.value].copy(), pubs[: npub.value].copy(), list(timing))


Engine.gens_derive = _gens_derive
Engine.gens_upload = _gens_upload
Engine.gens_download = _gens_download
Engine.prove_scenario = _prove_scenario


class HostTranscript:
    """The product's merlin::Transcript (host C++), for tests and for driving bp_ipa_create."""

    def __init__(self, label):
        self.h = C.c_void_p(lib().bp_transcript_new(bytes(label), C.c_size_t(len(label))))

    def __del__(self):
        try:
            lib().bp_transcript_free(self.h)
        except Exception:
            pass

    def append_message(self, label, msg):
        lib().bp_transcript_append_message(self.h, bytes(label), bytes(msg), C.c_size_t(len(msg)))

    def append_u64(self, label, x):
        self.append_message(label, int(x).to_bytes(8, "little"))

    def challenge_bytes(self, label, n):
        out = C.create_string_buffer(n)
        lib().bp_transcript_challenge_bytes(self.h, bytes(label), out, C.c_size_t(n))
        return out.raw

    def append_point(self, curve, label, xy):
        check(lib().bp_transcript_append_point(curve, self.h, bytes(label), ptr(np.ascontiguousarray(xy, dtype=np.uint64))), "append_point")

    def challenge_scalar(self, curve, label):
        out = np.zeros(4, dtype=np.uint64)
        check(lib().bp_transcript_challenge_scalar(curve, self.h, bytes(label), ptr(out)), "challenge_scalar")
        return out


def debug_rng_draws(curve, transcript, witness, seeds, count):
    """seeds: one 32-byte seed (scalar path) or eight (AVX-512 x8 path); returns (lanes, count, 4) u64 or None if unavailable"""
    w = np.ascontiguousarray(witness, dtype=np.uint64).reshape(-1, 4)
    lanes = len(seeds) // 32
    out = np.zeros((lanes, count, 4), dtype=np.uint64)
    rc = lib().bp_debug_rng_draws(curve, transcript.h, ptr(w), C.c_size_t(len(w)), bytes(seeds), lanes, C.c_size_t(count), ptr(out))
    if rc == _lib.BP_E_ARG and lanes == 8:
        return None
    check(rc, "bp_debug_rng_draws")
    return out


def debug_append_points_x8(curve, transcripts, label, points):
    """points: (lanes, npts, 8) u64; appends them to the HostTranscripts in lockstep (AVX-512 Keccak-f x8); False if unavailable"""
    pts = np.ascontiguousarray(points, dtype=np.uint64)
    lanes, npts = pts.shape[0], pts.shape[1]
    arr = (C.c_void_p * lanes)(*[t.h for t in transcripts])
    rc = lib().bp_debug_append_points_x8(curve, arr, lanes, bytes(label), ptr(pts), C.c_size_t(npts))
    if rc == _lib.BP_E_ARG:
        return False
    check(rc, "bp_debug_append_points_x8")
    return True


def debug_challenge_x8(transcripts, msg_label, msg, chal_label, nbytes):
    """eight HostTranscripts at the same STROBE position: append `msg` under `msg_label` and draw `nbytes` challenge bytes each under
    `chal_label`, all in lockstep (AVX-512 Keccak-f x8); returns an (8, nbytes) u8 array, or None if unavailable"""
    arr = (C.c_void_p * 8)(*[t.h for t in transcripts])
    out = np.zeros((8, nbytes), dtype=np.uint8)
    m = np.frombuffer(bytes(msg), dtype=np.uint8).copy() if len(msg) else np.zeros(1, dtype=np.uint8)
    rc = lib().bp_debug_challenge_x8(arr, bytes(msg_label), ptr(m), C.c_size_t(len(msg)), bytes(chal_label), C.c_size_t(nbytes), ptr(out))
    if rc == _lib.BP_E_ARG:
        return None
    check(rc, "bp_debug_challenge_x8")
    return out


def pedersen_gens(curve):
    B, Bb = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    check(lib().bp_pedersen_gens(curve, ptr(B), ptr(Bb)), "bp_pedersen_gens")
    return B, Bb


def host_derive_generators(curve, which_H, party, count):
    out = np.zeros((count, 8), dtype=np.uint64)
    check(lib().bp_host_derive_generators(curve, int(which_H), party, C.c_size_t(count), ptr(out)), "bp_host_derive_generators")
    return out


def host_sha3_512(msg):
    out = C.create_string_buffer(64)
    lib().bp_host_sha3_512(bytes(msg), C.c_size_t(len(msg)), out)
    return out.raw


# ---- verifier ----------------------------------------------------------------------------------------
def _prm(params):
    prm = np.zeros(8, dtype=np.uint64)
    prm[: len(params)] = np.array(params, dtype=np.uint64)
    return prm


def _verify_scenario(self, scenario, params, proof, commitments, publics):
    """Verifier::verify for a scenario statement; returns the C status (0 = Ok, -4 VerificationError, -6 FormatError, ...)"""
    cm = np.ascontiguousarray(commitments, dtype=np.uint64).reshape(-1, 8)
    pb = np.ascontiguousarray(publics, dtype=np.uint64).reshape(-1, 4)
    return lib().bp_r1cs_verify_scenario(self.ctx, scenario, ptr(_prm(params)), bytes(proof), C.c_size_t(len(proof)), ptr(cm), C.c_size_t(len(cm)),
                                         ptr(pb), C.c_size_t(len(pb)))


class PackedInstances:
    """The flat arrays bp_r1cs_batch_verify_scenarios takes (what a caller in the reference's language would hand over directly),
    built once from a list of (scenario, params, proof_bytes, commitments, publics)."""

    def __init__(self, instances):
        n = len(instances)
        self.n = n
        if n == 0:   # an empty batch is valid (the reference's mega-check of nothing is the identity)
            self.scen = (C.c_int * 1)()
            self.prm = np.zeros(8, dtype=np.uint64)
            self.proofs = b""
            self.plens = (C.c_size_t * 1)()
            self.cms = np.zeros((1, 8), dtype=np.uint64)
            self.ms = (C.c_size_t * 1)()
            self.pubs = np.zeros((1, 4), dtype=np.uint64)
            self.npubs = (C.c_size_t * 1)()
            return
        self.scen = (C.c_int * n)(*[i[0] for i in instances])
        self.prm = np.concatenate([_prm(i[1]) for i in instances])
        self.proofs = b"".join(i[2] for i in instances)
        self.plens = (C.c_size_t * n)(*[len(i[2]) for i in instances])
        cm_l = [np.asarray(i[3], dtype=np.uint64).reshape(-1, 8) for i in instances]
        self.cms = np.ascontiguousarray(np.concatenate(cm_l))
        self.ms = (C.c_size_t * n)(*[len(c) for c in cm_l])
        pubs_l = [np.asarray(i[4], dtype=np.uint64).reshape(-1, 4) for i in instances]
        self.pubs = np.ascontiguousarray(np.concatenate(pubs_l + [np.zeros((1, 4), dtype=np.uint64)]))
        self.npubs = (C.c_size_t * n)(*[len(p) for p in pubs_l])


def pack_instances(instances):
    return PackedInstances(instances)


def _batch_verify(self, instances, alpha_seed, alpha_skip=0, want_point=False):
    """batch_verify; instances: list of (scenario, params, proof_bytes, commitments, publics) or a PackedInstances.  Returns
    (status, timing[5]) or (status, timing, check_point) when want_point (proof-sharded multi-GPU use, see parallel.py)."""
    pk = instances if isinstance(instances, PackedInstances) else PackedInstances(instances)
    timing = (C.c_double * 5)()
    pt = np.zeros(8, dtype=np.uint64)
    rc = lib().bp_r1cs_batch_verify_scenarios(self.ctx, C.c_size_t(pk.n), pk.scen, ptr(pk.prm), pk.proofs, pk.plens, ptr(pk.cms), pk.ms, ptr(pk.pubs), pk.npubs,
                                              bytes(alpha_seed), timing, C.c_size_t(alpha_skip), ptr(pt))
    if want_point:
        return rc, list(timing), pt
    return rc, list(timing)


def _verification_gh(self, n1, wL, wR, wO, y, x, u, a, b, ipa_challenges):
    """g_scalars / h_scalars of Verifier::verification_scalars (canonical integers, 2^k each) from the caller's flattened vectors"""
    wL, wR, wO = u64arr(wL, 4), u64arr(wR, 4), u64arr(wO, 4)
    ch = u64arr(ipa_challenges, 4)
    k = len(ch)
    N = 1 << k
    g, h = np.zeros((N, 4), dtype=np.uint64), np.zeros((N, 4), dtype=np.uint64)
    sc = [np.ascontiguousarray(v, dtype=np.uint64).reshape(4) for v in (y, x, u, a, b)]
    check(lib().bp_r1cs_verification_gh(self.ctx, C.c_size_t(len(wL)), C.c_size_t(n1), ptr(wL), ptr(wR), ptr(wO), *[ptr(v) for v in sc], ptr(ch),
                                        C.c_size_t(k), ptr(g), ptr(h)), "bp_r1cs_verification_gh")
    return g, h


Engine.verification_gh = _verification_gh
Engine.verify_scenario = _verify_scenario
Engine.batch_verify = _batch_verify


def host_points_sum(curve, pts_xy):
    pts = u64arr(pts_xy, 8)
    out = np.zeros(8, dtype=np.uint64)
    check(lib().bp_host_points_sum(curve, ptr(pts), C.c_size_t(len(pts)), ptr(out)), "bp_host_points_sum")
    return out


# ---- statements (setup separated from prove(), several proofs in flight) -----------------------------------
class Statement:
    """Prover::new + commits + gadget for a scenario.  Host only by default; with `engine` the statement's Pedersen
    commitments are one GPU batch (same statement either way).  prove(engine) consumes it."""

    def __init__(self, curve, scenario, params, seed, engine=None):
        self.h = C.c_void_p()
        if engine is not None:
            if engine.curve != curve:
                raise ValueError("Statement: engine curve differs")
            check(lib().bp_stmt_prover_create_dev(engine.ctx, scenario, ptr(_prm(params)), bytes(seed), C.byref(self.h)), "bp_stmt_prover_create_dev")
        else:
            check(lib().bp_stmt_prover_create(curve, scenario, ptr(_prm(params)), bytes(seed), C.byref(self.h)), "bp_stmt_prover_create")
        self.curve = curve

    def info(self, m_cap=1 << 16):
        commits = np.zeros((m_cap, 8), dtype=np.uint64)
        pubs = np.zeros((8, 4), dtype=np.uint64)
        m, npub, nm, nq = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        check(lib().bp_stmt_info(self.h, ptr(commits), C.c_size_t(m_cap), C.byref(m), ptr(pubs), C.byref(npub), C.byref(nm), C.byref(nq)), "bp_stmt_info")
        return commits[: m.value].copy(), pubs[: npub.value].copy(), nm.value, nq.value

    def precompute(self):
        """host-only head of prove(): the TranscriptRng chain (needs no GPU)"""
        check(lib().bp_stmt_precompute(self.h), "bp_stmt_precompute")

    def prove(self, eng):
        buf = C.create_string_buffer(1 << 16)
        plen = C.c_size_t(len(buf))
        timing = (C.c_double * 8)()
        check(lib().bp_stmt_prove(eng.ctx, self.h, buf, C.byref(plen), timing), "bp_stmt_prove")
        return buf.raw[: plen.value], list(timing)

    def free(self):
        if self.h:
            lib().bp_stmt_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _pedersen_commit_batch(self, v, blind):
    """PedersenGens::commit for m (value, blinding) pairs (ark words, m x 4 each) -> m x 8 affine words"""
    v, blind = u64arr(v, 4), u64arr(blind, 4)
    if v.shape != blind.shape:
        raise ValueError("pedersen_commit_batch: length mismatch")
    out = np.zeros((v.shape[0], 8), dtype=np.uint64)
    check(lib().bp_pedersen_commit_batch(self.ctx, ptr(v), ptr(blind), C.c_size_t(v.shape[0]), ptr(out)), "bp_pedersen_commit_batch")
    return out


def _share_gens_from(self, other):
    check(lib().bp_gens_share(self.ctx, other.ctx), "bp_gens_share")
    self.gens_capacity = getattr(other, "gens_capacity", 0)
    self._gens_owner = other  # keep the owner alive


Engine.share_gens_from = _share_gens_from
Engine.pedersen_commit_batch = _pedersen_commit_batch


def precompute_batch(stmts):
    """host-only head of prove() for many statements; groups of 8 same-shaped ones share one AVX-512 Keccak-f x8 stream"""
    arr = (C.c_void_p * len(stmts))(*[s.h for s in stmts])
    check(lib().bp_stmt_precompute_batch(arr, C.c_size_t(len(stmts))), "bp_stmt_precompute_batch")


def _ipa_verify(self, n, G_factors, H_factors, P, Q, G_vec, H_vec, L_vec, R_vec, challenges, a, b):
    """InnerProductProof::verify; challenges = the u_j the caller's transcript replay produced.  Returns the C status."""
    Lv = np.ascontiguousarray(L_vec, dtype=np.uint64).reshape(-1, 8)
    Rv = np.ascontiguousarray(R_vec, dtype=np.uint64).reshape(-1, 8)
    ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(-1, 4)
    k = len(Lv)
    pad8, pad4 = np.zeros((1, 8), dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64)
    arrs = [u64arr(G_factors, 4), u64arr(H_factors, 4), u64arr(P, 8), u64arr(Q, 8), u64arr(G_vec, 8), u64arr(H_vec, 8)]
    return lib().bp_ipa_verify(self.ctx, C.c_size_t(n), *[ptr(x) for x in arrs], ptr(Lv if k else pad8), ptr(Rv if k else pad8), C.c_size_t(k),
                               ptr(ch if k else pad4), ptr(u64arr(a, 4)), ptr(u64arr(b, 4)))


Engine.ipa_verify = _ipa_verify


def msm_window_count(curve, n):
    """(windows, window_bits) of the engine's Pippenger schedule for an n-term MSM; depends only on (curve, n)"""
    w, c = C.c_int(0), C.c_int(0)
    check(lib().bp_msm_window_count(curve, C.c_size_t(n), C.byref(w), C.byref(c)), "bp_msm_window_count")
    return w.value, c.value


def _msm_dev_windows(self, d_bases, d_scalars, n, w_lo, w_hi, canonical=False):
    out = np.zeros(8, dtype=np.uint64)
    check(lib().bp_msm_dev_windows(self.ctx, d_bases.ptr, d_scalars.ptr, C.c_size_t(n), int(canonical), int(w_lo), int(w_hi), ptr(out)), "bp_msm_dev_windows")
    return out


Engine.msm_dev_windows = _msm_dev_windows


def _debug_decompress(self, compressed):
    """compressed: bytes of n x 33; returns (points (n,8) u64, ok (n,) u32)"""
    n = len(compressed) // 33
    out = np.zeros((n, 8), dtype=np.uint64)
    ok = np.zeros(n, dtype=np.uint32)
    check(lib().bp_debug_decompress(self.ctx, bytes(compressed), C.c_size_t(n), ptr(out), ptr(ok)), "bp_debug_decompress")
    return out, ok


Engine.debug_decompress = _debug_decompress


# ---- r1cs::ConstraintSystem / Prover / Verifier for the caller's own gadgets (include/arkbp.h "bp_cs") ----------------------
# Variables are (kind, index) pairs; a linear combination is a list of (variable, coefficient) with coefficients as 4 x u64
# Montgomery words — the shapes the reference's `Variable` and `LinearCombination` have (src/r1cs/linear_combination.rs).
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


class _ConstraintSystem:
    """What `impl ConstraintSystem for Prover / Verifier` offers (src/r1cs/constraint_system.rs:19-135), over a bp_cs handle."""

    def __init__(self, curve, transcript, handle=None):
        self.curve, self.transcript_obj = curve, transcript   # the transcript is borrowed: keep it alive
        self.h = handle if handle is not None else C.c_void_p()
        self._cbs = []
        self._cb_errors = []

    def free(self):
        if self.h:
            lib().bp_cs_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def transcript(self):
        return self.transcript_obj

    def multiply(self, left, right):
        lv, lc, nl = _lc_arrays(left)
        rv, rc, nr = _lc_arrays(right)
        out = np.zeros((3, 2), dtype=np.uint32)
        check(lib().bp_cs_multiply(self.h, ptr(lv), ptr(lc), C.c_size_t(nl), ptr(rv), ptr(rc), C.c_size_t(nr), ptr(out)), "bp_cs_multiply")
        return _vars_out(out)

    def allocate(self, assignment=None):
        out = np.zeros((1, 2), dtype=np.uint32)
        a = None if assignment is None else np.ascontiguousarray(assignment, dtype=np.uint64).reshape(4)
        check(lib().bp_cs_allocate(self.h, ptr(a) if a is not None else None, ptr(out)), "bp_cs_allocate")
        return _vars_out(out)[0]

    def allocate_multiplier(self, assignments=None):
        out = np.zeros((3, 2), dtype=np.uint32)
        if assignments is None:
            check(lib().bp_cs_allocate_multiplier(self.h, None, None, ptr(out)), "bp_cs_allocate_multiplier")
        else:
            l, r = [np.ascontiguousarray(x, dtype=np.uint64).reshape(4) for x in assignments]
            check(lib().bp_cs_allocate_multiplier(self.h, ptr(l), ptr(r), ptr(out)), "bp_cs_allocate_multiplier")
        return _vars_out(out)

    def constrain(self, lc):
        v, c, n = _lc_arrays(lc)
        check(lib().bp_cs_constrain(self.h, ptr(v), ptr(c), C.c_size_t(n)), "bp_cs_constrain")

    def allocate_multipliers(self, left=None, right=None, count=None):
        """bulk allocate_multiplier: returns the index of the first new multiplier"""
        first = C.c_uint32(0)
        if left is None:
            check(lib().bp_cs_allocate_multipliers(self.h, None, None, C.c_size_t(count), C.byref(first)), "bp_cs_allocate_multipliers")
        else:
            l, r = u64arr(left, 4), u64arr(right, 4)
            check(lib().bp_cs_allocate_multipliers(self.h, ptr(l), ptr(r), C.c_size_t(len(l)), C.byref(first)), "bp_cs_allocate_multipliers")
        return first.value

    def constrain_many(self, vars_, coefs, offsets):
        """bulk constrain in CSR form: constraint q owns terms [offsets[q], offsets[q+1])"""
        v = np.ascontiguousarray(vars_, dtype=np.uint32).reshape(-1, 2)
        c = u64arr(coefs, 4)
        off = (C.c_size_t * len(offsets))(*[int(x) for x in offsets])
        check(lib().bp_cs_constrain_many(self.h, ptr(v), ptr(c), off, C.c_size_t(len(offsets) - 1)), "bp_cs_constrain_many")

    def specify_randomized_constraints(self, fn):
        """fn(cs) runs in the randomized phase of prove / verify; cs.challenge_scalar(label) is available inside it"""
        me = self

        def thunk(_user, handle):
            try:
                view = me if handle == me.h.value else _ConstraintSystem(me.curve, None, C.c_void_p(handle))
                try:
                    fn(view)
                finally:
                    if view is not me:
                        view.h = None   # a borrowed handle (a like-instance of this gadget): not ours to free
                return 0
            except Exception as e:  # never unwind through C
                me._cb_errors.append(e)
                return -100

        cb = _RANDOMIZE_CB(thunk)
        self._cbs.append(cb)
        check(lib().bp_cs_specify_randomized_constraints(self.h, cb, None), "bp_cs_specify_randomized_constraints")

    def challenge_scalar(self, label):
        out = np.zeros(4, dtype=np.uint64)
        check(lib().bp_cs_challenge_scalar(self.h, bytes(label) + b"\0", ptr(out)), "bp_cs_challenge_scalar")
        return out

    def metrics(self):
        a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        check(lib().bp_cs_metrics(self.h, C.byref(a), C.byref(b), C.byref(c)), "bp_cs_metrics")
        return a.value, b.value, c.value


class ProverCS(_ConstraintSystem):
    """r1cs::Prover (src/r1cs/prover.rs): `ProverCS(curve, transcript)` = Prover::new(&pc_gens, transcript)"""

    def __init__(self, curve, transcript):
        super().__init__(curve, transcript)
        check(lib().bp_prover_new(curve, transcript.h, C.byref(self.h)), "bp_prover_new")

    def commit(self, v, v_blinding, engine=None):
        """Prover::commit for one value or for arrays of values: returns (V points (count, 8), variables)"""
        v, b = u64arr(v, 4), u64arr(v_blinding, 4)
        V = np.zeros((len(v), 8), dtype=np.uint64)
        vars_ = np.zeros((len(v), 2), dtype=np.uint32)
        check(lib().bp_prover_commit(self.h, engine.ctx if engine is not None else None, ptr(v), ptr(b), C.c_size_t(len(v)), ptr(V), ptr(vars_)), "bp_prover_commit")
        return V, _vars_out(vars_)

    def prove(self, engine, rng_bytes):
        """Prover::prove(prng, &bp_gens): rng_bytes = the 32 bytes the external prng yields; returns R1CSProof::to_bytes()"""
        buf = C.create_string_buffer(1 << 16)
        plen = C.c_size_t(len(buf))
        rc = lib().bp_prover_prove(engine.ctx, self.h, bytes(rng_bytes), buf, C.byref(plen), None)
        if self._cb_errors:
            raise self._cb_errors[0]
        check(rc, "bp_prover_prove")
        return buf.raw[: plen.value]


class VerifierCS(_ConstraintSystem):
    """r1cs::Verifier (src/r1cs/verifier.rs): `VerifierCS(curve, transcript)` = Verifier::new(transcript);
    `VerifierCS(curve, transcript, like=v0)` = the next instance of v0's gadget without recording it again."""

    def __init__(self, curve, transcript, like=None):
        super().__init__(curve, transcript)
        if like is None:
            check(lib().bp_verifier_new(curve, transcript.h, C.byref(self.h)), "bp_verifier_new")
        else:
            check(lib().bp_verifier_new_like(like.h, transcript.h, C.byref(self.h)), "bp_verifier_new_like")
            self._like = like   # its callbacks (ctypes thunks) serve this instance too

    def commit(self, V):
        V = u64arr(V, 8)
        vars_ = np.zeros((len(V), 2), dtype=np.uint32)
        check(lib().bp_verifier_commit(self.h, ptr(V), C.c_size_t(len(V)), ptr(vars_)), "bp_verifier_commit")
        return _vars_out(vars_)

    def verify(self, engine, proof):
        """Verifier::verify(&proof, &pc_gens, &bp_gens): returns the C status (0 = Ok(()))"""
        return lib().bp_verifier_verify(engine.ctx, self.h, bytes(proof), C.c_size_t(len(proof)))


def batch_verify_cs(engine, verifiers, proofs, alphas, want_point=False):
    """batch_verify(prng, instances, ..) over VerifierCS objects; alphas: (count, 4) words, the per-instance weights the
    reference draws from its prng (src/r1cs/verifier.rs:649) — required: equal weights would let the errors of two invalid
    proofs cancel.  Only a batch of ONE instance (Verifier::verify) may pass None.
    Returns the status, or (status, mega-check point) with want_point."""
    n = len(verifiers)
    if alphas is None and n > 1:
        raise ValueError("batch_verify_cs: per-instance weights (alphas) are required for more than one instance")
    hs = (C.c_void_p * max(n, 1))(*[v.h for v in verifiers])
    blob = b"".join(bytes(p) for p in proofs)
    lens = (C.c_size_t * max(n, 1))(*[len(p) for p in proofs])
    al = None if alphas is None else u64arr(alphas, 4)
    pt = np.zeros(8, dtype=np.uint64)
    rc = lib().bp_r1cs_batch_verify(engine.ctx, C.c_size_t(n), hs, blob, lens, ptr(al) if al is not None else None, None, ptr(pt))
    for v in verifiers:
        if v._cb_errors:
            raise v._cb_errors[0]
        src = getattr(v, "_like", None)
        if src is not None and src._cb_errors:
            raise src._cb_errors[0]
    return (rc, pt) if want_point else rc


def transcript_state(t):
    out = C.create_string_buffer(203)
    check(lib().bp_transcript_export_state(t.h, out), "bp_transcript_export_state")
    return out.raw


def transcript_from_state(state):
    t = HostTranscript(b"")
    check(lib().bp_transcript_import_state(t.h, bytes(state)), "bp_transcript_import_state")
    return t


def _debug_exp_iter(self, x, n):
    out = np.zeros((n, 4), dtype=np.uint64)
    check(lib().bp_debug_exp_iter(self.ctx, ptr(np.ascontiguousarray(x, dtype=np.uint64).reshape(4)), C.c_size_t(n), ptr(out)), "bp_debug_exp_iter")
    return out


def _debug_inner_product(self, a, b):
    a, b = u64arr(a, 4), u64arr(b, 4)
    out = np.zeros(4, dtype=np.uint64)
    check(lib().bp_debug_inner_product(self.ctx, ptr(a), ptr(b), C.c_size_t(len(a)), ptr(out)), "bp_debug_inner_product")
    return out


Engine.debug_exp_iter = _debug_exp_iter
Engine.debug_inner_product = _debug_inner_product


def _gens_fold_tables(self, count, window_bits=0, budget_bytes=0, rank=0, world=1):
    """fixed-base tables of G[0..count), H[0..count) for the first fold round(s) of the prover (bp_gens_fold_tables); with
    world > 1 only this rank's slice of them (the generators rank + i * world: bp_gens_fold_tables_slice) — what the index-cyclic
    inner-product argument of a sharded prover looks up.  Returns (window bits chosen, table bytes)"""
    w, nbytes = C.c_int(0), C.c_size_t(0)
    if world > 1:
        check(lib().bp_gens_fold_tables_slice(self.ctx, C.c_size_t(count), int(window_bits), C.c_size_t(budget_bytes), int(rank), int(world), C.byref(w), C.byref(nbytes)),
              "bp_gens_fold_tables_slice")
    else:
        check(lib().bp_gens_fold_tables(self.ctx, C.c_size_t(count), int(window_bits), C.c_size_t(budget_bytes), C.byref(w), C.byref(nbytes)), "bp_gens_fold_tables")
    return w.value, nbytes.value


Engine.gens_fold_tables = _gens_fold_tables


def _gens_msm_tables(self, count):
    """fixed-base rows of G[0..count), H[0..count) and PedersenGens for the prover's MSMs over the generator tables
    (bp_gens_msm_tables); returns the table bytes"""
    nbytes = C.c_size_t(0)
    check(lib().bp_gens_msm_tables(self.ctx, C.c_size_t(count), C.byref(nbytes)), "bp_gens_msm_tables")
    return nbytes.value


Engine.gens_msm_tables = _gens_msm_tables


def _gens_tables_check(self):
    """bp_gens_tables_check: (fold-table entries, fixed-base MSM rows) that break the chain rule — (0, 0) for sound tables"""
    a, b = C.c_uint64(0), C.c_uint64(0)
    check(lib().bp_gens_tables_check(self.ctx, C.byref(a), C.byref(b)), "bp_gens_tables_check")
    return a.value, b.value


def _debug_tables_ptr(self, which):
    p, n = C.c_void_p(), C.c_size_t(0)
    check(lib().bp_debug_tables_ptr(self.ctx, int(which), C.byref(p), C.byref(n)), "bp_debug_tables_ptr")
    return p.value, n.value


def _debug_poke(self, dptr, data=None, nbytes=0):
    """reads (data is None) or writes raw device bytes at an address of this ctx's device (tests corrupt a table entry with it)"""
    if data is None:
        out = np.zeros(nbytes, dtype=np.uint8)
        check(lib().bp_dev_download(self.ctx, ptr(out), C.c_void_p(dptr), C.c_size_t(nbytes)), "bp_dev_download")
        return out
    arr = np.ascontiguousarray(data, dtype=np.uint8)
    check(lib().bp_dev_upload(self.ctx, C.c_void_p(dptr), ptr(arr), C.c_size_t(arr.nbytes)), "bp_dev_upload")


Engine.gens_tables_check = _gens_tables_check
Engine.debug_tables_ptr = _debug_tables_ptr
Engine.debug_poke = _debug_poke


# ---- native collectives (RCCL inside the library) -------------------------------------------------------------------------
def rccl_unique_id():
    """ncclGetUniqueId (128 bytes): one rank makes it, the host's bootstrap hands it to the others"""
    out = C.create_string_buffer(128)
    check(lib().bp_rccl_unique_id(out), "bp_rccl_unique_id")
    return out.raw


def _rccl_init(self, unique_id, rank, world):
    """ncclCommInitRank on this ctx (collective): window-sharded mode with both exchanges as ncclAllGather on the ctx's stream"""
    check(lib().bp_ctx_rccl_init(self.ctx, bytes(unique_id), int(rank), int(world)), "bp_ctx_rccl_init")


def _rccl_shutdown(self):
    check(lib().bp_ctx_rccl_shutdown(self.ctx), "bp_ctx_rccl_shutdown")


def _collective_stats(self):
    n, s = C.c_uint64(0), C.c_double(0)
    check(lib().bp_ctx_collective_stats(self.ctx, C.byref(n), C.byref(s)), "bp_ctx_collective_stats")
    return n.value, s.value


def _debug_rccl_allgather(self, block, world):
    blk = np.ascontiguousarray(block, dtype=np.uint8).reshape(-1)
    out = np.zeros((world, blk.size), dtype=np.uint8)
    check(lib().bp_debug_rccl_allgather(self.ctx, ptr(blk), C.c_size_t(blk.size), ptr(out)), "bp_debug_rccl_allgather")
    return out


Engine.rccl_init = _rccl_init
Engine.rccl_shutdown = _rccl_shutdown
Engine.collective_stats = _collective_stats
Engine.debug_rccl_allgather = _debug_rccl_allgather


# ---- verifier front end on the device (csrc/vfe.hip): test hooks ----------------------------------------------------------
def vfe_schedule_replay(state203, absorb_commitments, m, k, n, items):
    """CPU run of the data-independent sponge schedule of one verification (bp_debug_vfe_schedule_replay).  items: (count, 72)
    uint8.  Returns ((6 + k, 32) uint8 challenge seeds, number of Keccak-f permutations)."""
    items = np.ascontiguousarray(items, dtype=np.uint8)
    seeds = np.zeros((6 + k, 32), dtype=np.uint8)
    nb = C.c_uint32(0)
    check(lib().bp_debug_vfe_schedule_replay(bytes(state203), int(bool(absorb_commitments)), C.c_uint64(m), C.c_uint32(k), C.c_uint64(n), ptr(items), ptr(seeds), C.byref(nb)),
          "bp_debug_vfe_schedule_replay")
    return seeds, nb.value


def _debug_vfe_challenges(self, proofs, commitments, states203, absorb_commitments):
    """k_vfe_points + k_vfe_sponge for same-shaped proofs.  proofs: list of equal-length bytes; commitments: (count, m, 8) uint64 ark
    words; states203: one 203-byte state (shared) or one per proof.  Returns (seeds (count, 6 + k, 32) uint8,
    challenges (count, 6 + k, 4) uint64 ark words, status)."""
    count, plen = len(proofs), len(proofs[0])
    assert all(len(p) == plen for p in proofs)
    k = (plen - 539) // 66
    V = np.ascontiguousarray(commitments, dtype=np.uint64).reshape(count, -1, 8)
    m = V.shape[1]
    shared = isinstance(states203, (bytes, bytearray))
    st = bytes(states203) if shared else b"".join(bytes(s) for s in states203)
    seeds = np.zeros((count, 6 + k, 32), dtype=np.uint8)
    chal = np.zeros((count, 6 + k, 4), dtype=np.uint64)
    status = C.c_uint32(0)
    blob = b"".join(proofs)
    check(lib().bp_debug_vfe_challenges(self.ctx, C.c_size_t(count), blob, C.c_size_t(plen), ptr(V) if m else None, C.c_size_t(m), st, int(shared), int(bool(absorb_commitments)),
                                        ptr(seeds), ptr(chal), C.byref(status)), "bp_debug_vfe_challenges")
    return seeds, chal, status.value


def _vfe_stats(self):
    a, b = C.c_uint64(0), C.c_uint64(0)
    check(lib().bp_ctx_vfe_stats(self.ctx, C.byref(a), C.byref(b)), "bp_ctx_vfe_stats")
    return a.value, b.value


Engine.debug_vfe_challenges = _debug_vfe_challenges
Engine.vfe_stats = _vfe_stats


def debug_verify_challenges(curve, instances, use_x8):
    """host only: the challenge sequences of up to eight scenario verifications (bp_debug_verify_challenges).  Returns a list of
    (nchal, 4) uint64 arrays, or None when the lockstep replay does not apply to the group."""
    pk = instances if isinstance(instances, PackedInstances) else PackedInstances(instances)
    out = np.zeros((pk.n, 40, 4), dtype=np.uint64)
    nch = (C.c_size_t * pk.n)()
    rc = lib().bp_debug_verify_challenges(curve, C.c_size_t(pk.n), pk.scen, ptr(pk.prm), pk.proofs, pk.plens, ptr(pk.cms), pk.ms, ptr(pk.pubs), pk.npubs, int(bool(use_x8)),
                                          ptr(out), nch)
    if rc == 1:
        return None
    check(rc, "bp_debug_verify_challenges")
    return [out[j, : nch[j]].copy() for j in range(pk.n)]


def _msm_stats(self):
    a, b = C.c_uint64(0), C.c_uint64(0)
    check(lib().bp_ctx_msm_stats(self.ctx, C.byref(a), C.byref(b)), "bp_ctx_msm_stats")
    return a.value, b.value


Engine.msm_stats = _msm_stats


def _direct_stats(self):
    """(MSMs answered from the direct window tables of the small-statement path, generators per vector the tables cover)"""
    a, b = C.c_uint64(0), C.c_size_t(0)
    check(lib().bp_ctx_direct_stats(self.ctx, C.byref(a), C.byref(b)), "bp_ctx_direct_stats")
    return a.value, b.value


Engine.direct_stats = _direct_stats


def _gens_direct_tables(self, count):
    """bp_gens_direct_tables: build (count > 0) or free (0) the direct window tables of the small-statement path; returns their size in bytes"""
    nbytes = C.c_size_t(0)
    check(lib().bp_gens_direct_tables(self.ctx, C.c_size_t(count), C.byref(nbytes)), "bp_gens_direct_tables")
    return nbytes.value


Engine.gens_direct_tables = _gens_direct_tables


def _fold_stats(self):
    """(first folds deferred, second folds produced straight from the fold tables) — the two-rounds-from-the-tables schedule"""
    a, b = C.c_uint64(0), C.c_uint64(0)
    check(lib().bp_ctx_fold_stats(self.ctx, C.byref(a), C.byref(b)), "bp_ctx_fold_stats")
    return a.value, b.value


Engine.fold_stats = _fold_stats
