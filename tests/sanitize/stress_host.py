#!/usr/bin/env python3
"""Host-layer stress for the sanitizer builds (tests/sanitize/Makefile): everything here runs WITHOUT a GPU, through the C ABI of a
libarkbp_hip.so whose host side was compiled with -fsanitize=address,undefined or -fsanitize=thread.

What it drives (the multi-threaded host code of the product; VERDICT r03 item 7, ADVICE r02's use-after-free / data race area):
  * batch verification's host side as a dry run on a device-less ctx (bp_debug_ctx_create_hostonly): framing, square roots on the
    pool, shared source recordings and like-instances, the live and the lockstep (AVX-512 x8) transcript replays, circuit templates,
    the template cache's eviction (> 64 structures), two-phase statements with their randomized closures, error paths (malformed
    proofs, an identity point, doomed batches), with 1 / 3 / 8 host threads — the staging checksum must not depend on the thread count;
  * the same through recorded handles (bp_verifier_new / bp_verifier_new_like over a caller gadget);
  * several batches at once from several caller threads on their own ctxs (as bench.py does);
  * the prover's host-only head: statement construction and the TranscriptRng x8 stage from concurrent threads.
The proofs come from the CPU oracle (test infrastructure)."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ark_bulletproofs_amd import engine as E   # noqa: E402
from oracle import pyoracle as O               # noqa: E402
import gadgets as GD                           # noqa: E402

GENS = 256
SEED = bytes([5]) * 32
TUNE_HOST_THREADS = 6
LABEL = b"GenericGadgetTest"


def proofs_for(cv, sc, prm, count, distinct, gens=GENS):
    base = []
    for i in range(distinct):
        pr = O.r1cs_prove(cv, sc, prm, bytes([40 + i]) * 32, gens, m_cap=64)
        assert pr.rc == 0
        base.append((sc, prm, pr.proof, pr.commitments, pr.publics))
    return [base[i % distinct] for i in range(count)]


def dry(eng, inst, seed=SEED):
    rc, _, pt = eng.batch_verify(inst, seed, want_point=True)
    return rc, int(pt[0]), int(pt[1]), int(pt[2])


def scenario_batches(cv):
    eng = E.Engine.host_only(cv, GENS)
    like = proofs_for(cv, O.SC_MULTI_RANGE, [3, 8, 0], 600, 4)                      # one shape, two 512-blocks: lockstep replay
    mixed = like[:40] + proofs_for(cv, O.SC_RANGE, [16, 77], 9, 2) + proofs_for(cv, O.SC_SHUFFLE, [4], 10, 2) + proofs_for(cv, O.SC_SQUARE_CHAIN, [20, 0], 7, 1) + like[40:90]
    many = []                                                                        # > 64 circuit structures: the template cache evicts
    for nbits in range(1, 65):
        many += proofs_for(cv, O.SC_RANGE, [nbits, 1], 2, 1)
    many += proofs_for(cv, O.SC_MULTI_RANGE, [2, 8, 0], 5, 1) + proofs_for(cv, O.SC_MULTI_RANGE, [4, 4, 0], 5, 1) + proofs_for(cv, O.SC_EXAMPLE, [3, 4, 6, 1, 40, 9], 3, 1)
    sums = {}
    for threads in (1, 3, 8):
        eng.set_tuning(TUNE_HOST_THREADS, threads)
        for name, inst in (("like", like), ("mixed", mixed), ("many", many), ("many-again", many[::-1][:70] + many[:70])):
            rc, cs, cnt, groups = dry(eng, inst)
            assert rc == 0 and cnt == len(inst), (name, rc, cnt)
            key = (name,)
            assert sums.setdefault(key, cs) == cs, "staging checksum depends on the thread count (%s, %d threads)" % (name, threads)
    # error paths: FormatError wins over a later VerificationError-class failure; an identity where validate_and_append_point refuses
    s, p, proof, cm, pb = like[7]
    bad = bytearray(proof); bad[32] = 0xC0
    inst = list(like[:30]); inst[12] = (s, p, bytes(bad), cm, pb)
    assert dry(eng, inst)[0] == -6
    bad = bytearray(proof); bad[6 * 33: 7 * 33] = b"\x00" * 32 + b"\x40"
    inst = list(like[:30]); inst[3] = (s, p, bytes(bad), cm, pb)
    assert dry(eng, inst)[0] == -4
    inst = list(like[:30]); inst[5] = (s, p, proof[:-1], cm, pb)                     # truncated
    assert dry(eng, inst)[0] == -6
    inst = list(like[:20]) + proofs_for(cv, O.SC_MULTI_RANGE, [40, 8, 0], 1, 1, gens=512)       # needs more generators than the ctx holds: doomed batch
    assert dry(eng, inst)[0] == -5
    eng.close()


def handle_batches(cv):
    """recorded verifiers + like-instances of a caller gadget, single- and two-phase"""
    eng = E.Engine.host_only(cv, GENS)
    F = GD.Field(O, cv)
    for two_phase in (False, True):
        kw = dict(n_mul=9, n_extra=2, n_alloc=1)
        items = []
        for w in range(3):
            vals, blinds = GD.make_witness(F, 70 + w, 2)
            p = O.ProverCS(cv, LABEL)
            p.transcript().append_message(b"dom-sep", b"generic gadget v1")
            p.start()
            V, vars_ = p.commit([F.w(v) for v in vals], [F.w(b) for b in blinds])
            wit = GD.Witness(F)
            for var, v in zip(vars_, vals):
                wit.val[var] = v
            publics = []
            GD.random_program(p, F, 31, wit, vars_, two_phase=two_phase, publics=publics, **kw)
            items.append((p.prove(GENS, bytes([w + 1]) * 32), V, publics))
        alphas = O.fe_rand(O.fid(cv, True), bytes([4]) * 32, 40)
        for threads in (1, 8):
            eng.set_tuning(TUNE_HOST_THREADS, threads)

            def mk(V, publics, like=None):
                t = E.HostTranscript(LABEL)
                t.append_message(b"dom-sep", b"generic gadget v1")
                v = E.VerifierCS(cv, t, like=like)
                vars_ = v.commit(V)
                if like is None:
                    GD.random_program(v, F, 31, None, vars_, two_phase=two_phase, publics=publics, **kw)
                return v

            v0 = mk(items[0][1], items[0][2])
            vs, pf = [v0], [items[0][0]]
            for k in range(1, 40):
                pr, V, pb = items[k % 3]
                vs.append(mk(V, pb, like=v0) if k % 4 else mk(V, pb))     # like-instances and own recordings mixed
                pf.append(pr)
            rc, pt = E.batch_verify_cs(eng, vs, pf, alphas, want_point=True)
            assert rc == 0 and int(pt[1]) == 40, rc
    eng.close()


def concurrent_batches(cv):
    inst = proofs_for(cv, O.SC_MULTI_RANGE, [3, 8, 0], 130, 3) + proofs_for(cv, O.SC_SHUFFLE, [4], 12, 2)
    out, errs = [None] * 4, []

    def worker(i):
        try:
            eng = E.Engine.host_only(cv, GENS)
            eng.set_tuning(TUNE_HOST_THREADS, 3)
            for _ in range(3):
                out[i] = dry(eng, inst)
            eng.close()
        except Exception as exc:   # noqa: BLE001
            errs.append(exc)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert all(o == out[0] and o[0] == 0 for o in out), out


def prover_host_head(cv):
    errs = []

    def worker(i):
        try:
            stmts = [E.Statement(cv, E.SC_SQUARE_CHAIN, [200, 0], bytes([i, k]) + bytes(30)) for k in range(16)]
            E.precompute_batch(stmts)
            one = E.Statement(cv, E.SC_MULTI_RANGE, [3, 8, 0], bytes([90 + i]) * 32)
            one.precompute()
            for s in stmts + [one]:
                s.free()
        except Exception as exc:   # noqa: BLE001
            errs.append(exc)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


def main():
    for cv in (0, 1):
        scenario_batches(cv)
        handle_batches(cv)
        concurrent_batches(cv)
        prover_host_head(cv)
    # the lockstep replay against the live one, the device front end's schedule interpreter, generator derivation on the pool
    inst = proofs_for(0, O.SC_MULTI_RANGE, [3, 8, 0], 8, 8)
    live, x8 = E.debug_verify_challenges(0, inst, False), E.debug_verify_challenges(0, inst, True)
    assert x8 is None or all((a == b).all() for a, b in zip(live, x8))
    E.host_derive_generators(0, 0, 0, 300)
    print("host stress ok")


if __name__ == "__main__":
    main()
