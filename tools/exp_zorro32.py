import sys, time
sys.path.insert(0, "/root/repo")
import ark_bulletproofs_amd as A
from ark_bulletproofs_amd import engine as E
from bench import statement_seed
for curve in (0, 1):
    eng = A.Engine(curve=curve)
    eng.gens_derive(2048)
    for k in (2, 4, 8, 16, 32, 64):
        seed = statement_seed(7, k)
        for _ in range(3):
            pr = eng.prove_scenario(E.SC_SHUFFLE, [k], seed, m_cap=2 * k + 8)
            assert eng.verify_scenario(E.SC_SHUFFLE, [k], pr.proof, pr.commitments, pr.publics) == 0
        ts = []
        for i in range(10):
            t0 = time.perf_counter()
            pr = eng.prove_scenario(E.SC_SHUFFLE, [k], seed, m_cap=2 * k + 8)
            ts.append((time.perf_counter() - t0) * 1e3)
        print("curve %d k=%d" % (curve, k), " ".join("%.2f" % t for t in ts), flush=True)
    eng.close()
