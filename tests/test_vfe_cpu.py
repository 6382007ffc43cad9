"""The verifier front end's sponge schedule (csrc/vfe_sched.hpp) without a GPU: the data-independent plan the device kernel
k_vfe_sponge follows — constant images, message pieces and squeeze points per Keccak-f permutation — run by its CPU interpreter
must produce, byte for byte, the challenge_bytes outputs of a live merlin transcript replaying the verifier's schedule
(src/r1cs/verifier.rs:403-460, 516-519; src/inner_product_proof.rs:266-280), here through the product's host transcript
AND through the independent Python STROBE model of tests/pystrobe.py."""
import numpy as np
import pytest

from ark_bulletproofs_amd import engine as E

import pystrobe


def live_replay(tr_append, tr_challenge, m, k, n, items, absorb):
    """the schedule of verify_prepare_t (r1cs_host.inc), message for message, on a live transcript"""
    it = iter(items)
    out = []

    def pt(label):
        tr_append(label, bytes(next(it)[:65]))

    if absorb:
        for _ in range(m):
            pt(b"V")
    rest = list(it)
    pts, scal = rest[: 11 + 2 * k], rest[11 + 2 * k:]
    it = iter(pts)
    tr_append(b"m", int(m).to_bytes(8, "little"))
    pt(b"A_I1"), pt(b"A_O1"), pt(b"S1")
    tr_append(b"dom-sep", b"r1cs-1phase")
    pt(b"A_I2"), pt(b"A_O2"), pt(b"S2")
    out.append(tr_challenge(b"y")), out.append(tr_challenge(b"z"))
    pt(b"T_1"), pt(b"T_3"), pt(b"T_4"), pt(b"T_5"), pt(b"T_6")
    out.append(tr_challenge(b"u")), out.append(tr_challenge(b"x"))
    for label, s in zip((b"t_x", b"t_x_blinding", b"e_blinding"), scal):
        tr_append(label, bytes(s[:32]))
    out.append(tr_challenge(b"w"))
    tr_append(b"dom-sep", b"ipp v1")
    tr_append(b"n", int(n).to_bytes(8, "little"))
    L, R = pts[11: 11 + k], pts[11 + k:]
    for i in range(k):
        tr_append(b"L", bytes(L[i][:65])), tr_append(b"R", bytes(R[i][:65]))
        out.append(tr_challenge(b"u"))
    out.append(tr_challenge(b"r"))   # (from a clone upstream; nothing follows here)
    return out


@pytest.mark.parametrize("m,k,absorb,prefix", [(0, 0, 1, 0), (1, 1, 1, 3), (4, 3, 1, 17), (5, 2, 0, 160), (256, 14, 1, 0), (256, 14, 0, 77), (33, 5, 1, 165), (2, 31, 1, 9)])
def test_schedule_equals_live_transcript(m, k, absorb, prefix):
    rng = np.random.default_rng(1000 * m + 10 * k + absorb)
    nitems = (m if absorb else 0) + 11 + 2 * k + 3
    items = rng.integers(0, 256, size=(nitems, 72), dtype=np.uint8)
    n = 1 << k
    label = b"vfe schedule test"
    t = E.HostTranscript(label)
    junk = bytes(rng.integers(0, 256, size=prefix, dtype=np.uint8))
    t.append_message(b"prefix", junk)          # moves the starting position around the rate block
    state = E.transcript_state(t)
    seeds, nblocks = E.vfe_schedule_replay(state, absorb, m, k, n, items)
    exp = live_replay(lambda l, msg: t.append_message(l, msg), lambda l: bytes(t.challenge_bytes(l, 32)), m, k, n, items, absorb)
    assert [bytes(s) for s in seeds] == exp
    # the independent Python model (pinned by merlin's published test vector in tests/test_host_logic.py)
    if m <= 33:
        pt = pystrobe.Transcript(label)
        pt.append_message(b"prefix", junk)
        exp2 = live_replay(lambda l, msg: pt.append_message(l, msg), lambda l: bytes(pt.challenge_bytes(l, 32)), m, k, n, items, absorb)
        assert exp2 == exp
    assert nblocks >= 6 + k


def test_schedule_rejects_bad_arguments():
    from ark_bulletproofs_amd import _lib
    import ctypes as C

    L = _lib.lib()
    st = bytearray(203)
    st[200] = 200   # a position outside the rate
    buf = (C.c_uint8 * 72)()
    out = (C.c_uint8 * 1024)()
    assert L.bp_debug_vfe_schedule_replay(bytes(st), 0, C.c_uint64(0), C.c_uint32(0), C.c_uint64(1), buf, out, None) == _lib.BP_E_ARG
    assert L.bp_debug_vfe_schedule_replay(bytes(203), 0, C.c_uint64(0), C.c_uint32(32), C.c_uint64(1), buf, out, None) == _lib.BP_E_ARG
