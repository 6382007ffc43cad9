"""Stress driver (not part of the tests): repeated batch verifications with sizes around the pipeline block boundaries, valid and
tampered, to shake out ordering bugs between the main stream, the decode stream and the pinned staging halves."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ark_bulletproofs_amd as A  # noqa: E402
from ark_bulletproofs_amd import engine as E  # noqa: E402

for cv in (0, 1):
    eng = A.Engine(curve=cv)
    eng.gens_derive(256)
    kinds = [(3, [100, 0]), (4, [2, 16, 0]), (1, [16, 99]), (3, [7, 0]), (0, [6])]
    distinct = []
    for i, (sc, prm) in enumerate(kinds):
        pr = eng.prove_scenario(sc, prm, bytes([90 + i]) * 32, m_cap=64)
        distinct.append((sc, prm, pr.proof, pr.commitments, pr.publics))
    bad = bytearray(distinct[0][2])
    bad[11 * 33 + 9] ^= 4
    bad_inst = (distinct[0][0], distinct[0][1], bytes(bad), distinct[0][3], distinct[0][4])
    fails = 0
    for rep in range(6):
        for n in (1, 2, 511, 512, 513, 1023, 1024, 1025, 1537, 2049):
            inst = [distinct[(i * 3 + rep) % len(distinct)] for i in range(n)]
            rc, _ = eng.batch_verify(E.pack_instances(inst), bytes([rep, n & 255]) + bytes(30))
            if rc != 0:
                print("curve", cv, "valid batch rejected", n, rep, rc)
                fails += 1
            pos = (n * 7 + rep) % n
            inst[pos] = bad_inst
            rc, _ = eng.batch_verify(E.pack_instances(inst), bytes([rep, n & 255]) + bytes(30))
            if rc != -4:
                print("curve", cv, "tampered batch accepted or wrong code", n, rep, pos, rc)
                fails += 1
    print("curve", cv, "stress done, failures:", fails)
    eng.close()
