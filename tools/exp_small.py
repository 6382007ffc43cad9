import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import ark_bulletproofs_amd as A
from ark_bulletproofs_amd import engine as E
eng = A.Engine(curve=0)
eng.gens_derive(2048)
names = ["total", "stmt", "rng", "upload", "commit_msm", "flatten", "poly", "ipa"]
for k in (2, 16, 128, 1024):
    seed = bytes([k & 255]) * 32
    for _ in range(3):
        pr = eng.prove_scenario(E.SC_SHUFFLE, [k], seed, m_cap=2 * k + 8)
    t0 = time.perf_counter()
    for _ in range(10):
        pr = eng.prove_scenario(E.SC_SHUFFLE, [k], seed, m_cap=2 * k + 8)
    dt = (time.perf_counter() - t0) / 10
    print("k=%4d wall %.2f ms  " % (k, dt * 1e3) + "  ".join("%s %.2f" % (n, t * 1e3) for n, t in zip(names, pr.timing)))
    t0 = time.perf_counter()
    for _ in range(10):
        rc = eng.verify_scenario(E.SC_SHUFFLE, [k], pr.proof, pr.commitments, pr.publics)
    print("       verify %.2f ms rc %d" % ((time.perf_counter() - t0) / 10 * 1e3, rc))
