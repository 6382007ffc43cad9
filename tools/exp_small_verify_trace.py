"""one k-shuffle verification under rocprofv3 --kernel-trace (tools/trace_tail.py lists the last verification's kernels)"""
import sys, time
sys.path.insert(0, "/root/repo")
import ark_bulletproofs_amd as A
from ark_bulletproofs_amd import engine as E
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eng = A.Engine(curve=0)
eng.gens_derive(2048)
seed = bytes([k & 255]) * 32
pr = eng.prove_scenario(E.SC_SHUFFLE, [k], seed, m_cap=2 * k + 8)
for _ in range(3):
    assert eng.verify_scenario(E.SC_SHUFFLE, [k], pr.proof, pr.commitments, pr.publics) == 0
time.sleep(0.05)
t0 = time.perf_counter()
rc = eng.verify_scenario(E.SC_SHUFFLE, [k], pr.proof, pr.commitments, pr.publics)
print("k=%d verify wall %.3f ms rc %d" % (k, (time.perf_counter() - t0) * 1e3, rc))
