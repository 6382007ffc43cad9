"""CPU unit test of the DEVICE arithmetic headers (csrc/fp29.cuh, csrc/ec.cuh are host+device code):
compiled with g++ -DARKBP_CHECK_BOUNDS so every limb/value contract is asserted, and compared with
Python integers.  Not a product path — nothing in the library routes work through this harness."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

import pymodel as M

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ark_bulletproofs_amd", "csrc")
R = 1 << 256


@pytest.fixture(scope="module")
def st():
    so = os.path.join(CSRC, "libfp29_selftest.so")
    srcs = [os.path.join(CSRC, f) for f in ("fp29_selftest.cpp", "fp29.cuh", "ec.cuh", "ecq.cuh", "glv.cuh", "arkbp_params.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DARKBP_CHECK_BOUNDS", "-Wno-unknown-pragmas", "-o", so, srcs[0]])
    return C.CDLL(so)


def words(x):
    return np.array([(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)


def unwords(w):
    return sum(int(v) << (32 * i) for i, v in enumerate(w))


def p_of(fid):
    c = M.CURVES[fid >> 1]
    return c["r"] if fid & 1 else c["q"]


def fe_op(st, fid, op, a, b=0):
    p = p_of(fid)
    out = np.zeros(8, dtype=np.uint32)
    A, B = words(a * R % p), words(b * R % p)
    st.fp29_fe_op(fid, op, A.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def from_mont(w, p):
    return unwords(w) * pow(R, -1, p) % p


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_field_ops(st, fid):
    p = p_of(fid)
    rnd = random.Random(fid)
    edge = [0, 1, 2, p - 1, p - 2, (1 << 255) % p, (1 << 232) % p, (1 << 29) - 1, (p - 1) // 2, pow(R, -1, p), (p - 19) % p]
    cases = [(a, b) for a in edge for b in edge] + [(rnd.randrange(p), rnd.randrange(p)) for _ in range(300)]
    for a, b in cases:
        assert from_mont(fe_op(st, fid, 0, a, b), p) == a * b % p
        assert from_mont(fe_op(st, fid, 1, a, b), p) == (a + b) % p
        assert from_mont(fe_op(st, fid, 2, a, b), p) == (a - b) % p
        assert from_mont(fe_op(st, fid, 3, a), p) == a * a % p
        assert from_mont(fe_op(st, fid, 5, a), p) == a
        assert unwords(fe_op(st, fid, 6, a)) == a
        assert from_mont(fe_op(st, fid, 7, a), p) == (-a) % p
        assert from_mont(fe_op(st, fid, 8, a, b), p) == (a + b) * 2 * a % p
        assert from_mont(fe_op(st, fid, 9, a, b), p) == (a - 9 * b) % p
        assert from_mont(fe_op(st, fid, 11, a), p) == a
        assert int(fe_op(st, fid, 12, a, b)[0]) == (1 if a == b else 0)
        assert from_mont(fe_op(st, fid, 13, a, b), p) == (a * b + (a + b) * b) % p          # fe_mul2: two products, one reduction
        assert from_mont(fe_op(st, fid, 14, a, b), p) == (2 * a * b + (b - a) * (a + 2 * b)) % p
    for a in edge[1:] + [rnd.randrange(1, p) for _ in range(10)]:
        assert from_mont(fe_op(st, fid, 4, a), p) == pow(a, -1, p)
    # load_canon: words are a canonical integer
    a = rnd.randrange(p)
    out = np.zeros(8, dtype=np.uint32)
    A = words(a)
    st.fp29_fe_op(fid, 10, A.ctypes.data_as(C.c_void_p), A.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert from_mont(out, p) == a


def pt_words(cv, P):
    q = M.CURVES[cv]["q"]
    if P is None:
        return np.zeros(16, dtype=np.uint32)
    return np.concatenate([words(P[0] * R % q), words(P[1] * R % q)])


def pt_op(st, cv, op, P, Q, k=0):
    q = M.CURVES[cv]["q"]
    out = np.zeros(16, dtype=np.uint32)
    a, b, kk = pt_words(cv, P), pt_words(cv, Q), words(k)
    st.fp29_pt_op(cv, op, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), kk.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    if not out.any():
        return None
    return from_mont(out[:8], q), from_mont(out[8:], q)


@pytest.mark.parametrize("cv", [0, 1])
def test_group_ops(st, cv):
    c = M.CURVES[cv]
    G = (c["gx"], c["gy"])
    rnd = random.Random(10 + cv)
    pts = [M.mul(cv, G, rnd.randrange(1, c["r"])) for _ in range(6)]
    neg = lambda P: None if P is None else (P[0], (-P[1]) % c["q"])
    for P in pts[:4]:
        for Q in pts[:4] + [None, P, neg(P)]:
            assert pt_op(st, cv, 0, P, Q) == M.add(cv, P, Q)
            assert pt_op(st, cv, 1, P, Q) == M.add(cv, P, Q)
            assert pt_op(st, cv, 4, P, Q) == M.add(cv, M.add(cv, P, P), neg(Q))
        assert pt_op(st, cv, 0, None, P) == P and pt_op(st, cv, 1, None, P) == P
        assert pt_op(st, cv, 2, P, None) == M.add(cv, P, P)
        assert pt_op(st, cv, 5, P, None) == neg(P)
    assert pt_op(st, cv, 2, None, None) is None and pt_op(st, cv, 0, None, None) is None
    for k in [0, 1, 2, 3, c["r"] - 1, c["r"], rnd.randrange(c["r"]), rnd.randrange(c["r"]), (1 << 256) - 1]:
        for Q in [pts[1], None]:
            exp = M.add(cv, M.mul(cv, pts[0], k), M.add(cv, Q, Q))
            assert pt_op(st, cv, 3, pts[0], Q, k) == exp
    # 2*(k*P) + ... with Q chosen so the general adder hits its doubling / cancellation branches
    P = pts[2]
    assert pt_op(st, cv, 3, P, P, 2) == M.mul(cv, P, 4)
    assert pt_op(st, cv, 3, P, neg(P), 2) is None


@pytest.mark.parametrize("cv", [0, 1])
def test_quad_cooperative_schedules(st, cv):
    """ecq.cuh: Jacobian add / mixed add / doubling with the products of each dependency level dealt to the four lanes of a quad —
    every lane must end with the lane-per-operation result of ec.cuh, exceptional cases included (equal points: the doubling
    branch; opposite points; identity operands).  The DPP exchange is the CPU stand-in of ecq.cuh; bounds are asserted."""
    c = M.CURVES[cv]
    G = (c["gx"], c["gy"])
    rnd = random.Random(20 + cv)
    pts = [M.mul(cv, G, rnd.randrange(1, c["r"])) for _ in range(5)]
    neg = lambda P: None if P is None else (P[0], (-P[1]) % c["q"])   # noqa: E731
    out = np.zeros(16, dtype=np.uint32)
    cases = [(P, Q) for P in pts[:3] for Q in pts[:4]] + [(pts[0], None), (None, pts[1]), (None, None), (pts[2], neg(pts[2])), (pts[3], pts[3])]
    for P, Q in cases:
        a, b = pt_words(cv, P), pt_words(cv, Q)
        for op in (0, 1, 2):
            bad = st.fp29_quad_op(cv, op, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
            assert bad == 0, (P is None, Q is None, op, bad)
            P2 = M.add(cv, P, P)
            exp = M.add(cv, P2, Q) if op < 2 else M.add(cv, P2, P2)
            got = None if not out.any() else (from_mont(out[:8], c["q"]), from_mont(out[8:], c["q"]))
            assert got == exp


def test_glv_split_matches_big_integers(st):
    """glv.cuh: s = k1 + k2 * lambda (mod r) with |k1|, |k2| < 2^128 for secq256k1's scalar field — the word-level code against Python
    integers on extreme and random scalars"""
    r = M.CURVES[0]["r"]
    lam = next(l for l in (pow(g, (r - 1) // 3, r) for g in range(2, 20)) if l != 1)
    # the endomorphism's eigenvalue is one of the two primitive cube roots; the split must be consistent with ONE of them for all k
    rnd = random.Random(5)
    special = [0, 1, 2, r - 1, r - 2, (r - 1) // 2, (r + 1) // 2, lam, lam * lam % r, (1 << 128) - 1, 1 << 128, 1 << 255, 1 << 127, r // 3]
    lam_used = None
    worst = 0
    for k in special + [rnd.randrange(r) for _ in range(3000)]:
        kw = words(k)
        out = np.zeros(12, dtype=np.uint32)
        assert st.fp29_glv_split(kw.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 1
        m1 = sum(int(out[i]) << (32 * i) for i in range(4))
        m2 = sum(int(out[4 + i]) << (32 * i) for i in range(4))
        k1 = -m1 if out[8] & 1 else m1
        k2 = -m2 if out[8] & 2 else m2
        worst = max(worst, m1, m2)
        if lam_used is None and k2 != 0:
            lam_used = next(l for l in (lam, lam * lam % r) if (k1 + k2 * l - k) % r == 0)
        if lam_used is not None:
            assert (k1 + k2 * lam_used - k) % r == 0, k
    assert lam_used is not None and worst < (1 << 128) * 0.65
