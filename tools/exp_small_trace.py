"""one k-shuffle proof size under rocprofv3 --kernel-trace: tools/gpu_busy.py-style listing of one proof's kernels (start, duration, gap)"""
import sys, time
sys.path.insert(0, "/root/repo")
import ark_bulletproofs_amd as A
from ark_bulletproofs_amd import engine as E
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = A.Engine(curve=0)
eng.gens_derive(2048)
seed = bytes([k & 255]) * 32
for _ in range(3):
    pr = eng.prove_scenario(E.SC_SHUFFLE, [k], seed, m_cap=2 * k + 8)
time.sleep(0.05)
t0 = time.perf_counter()
pr = eng.prove_scenario(E.SC_SHUFFLE, [k], seed, m_cap=2 * k + 8)
print("k=%d wall %.3f ms" % (k, (time.perf_counter() - t0) * 1e3), ["%.3f" % (t * 1e3) for t in pr.timing])
