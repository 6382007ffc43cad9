"""prove latency (one proof at a time) with the direct window tables against the folding schedule, by padded size"""
import sys, time
sys.path.insert(0, "/root/repo")
import ark_bulletproofs_amd as A
from ark_bulletproofs_amd import engine as E
eng = A.Engine(curve=0)
top = int(sys.argv[1]) if len(sys.argv) > 1 else 16
eng.gens_derive(1 << top)
seed = bytes([5]) * 32
for logn in range(10, top + 1):
    N = 1 << logn
    out = []
    for direct in (1 << 16, 0):
        eng.set_tuning(12, direct)
        t_build = time.perf_counter()
        pr = eng.prove_scenario(E.SC_SQUARE_CHAIN, [N, 0], seed, m_cap=8)
        t_build = time.perf_counter() - t_build
        for _ in range(2):
            pr = eng.prove_scenario(E.SC_SQUARE_CHAIN, [N, 0], seed, m_cap=8)
        t0 = time.perf_counter()
        for _ in range(5):
            pr = eng.prove_scenario(E.SC_SQUARE_CHAIN, [N, 0], seed, m_cap=8)
        dt = (time.perf_counter() - t0) / 5
        out.append((dt, pr.timing[7], pr.proof, t_build))
    assert out[0][2] == out[1][2]
    print("N=2^%d  direct %.2f ms (ipa %.2f; first call %.0f ms)   folding %.2f ms (ipa %.2f)" % (logn, out[0][0] * 1e3, out[0][1] * 1e3, out[0][3] * 1e3, out[1][0] * 1e3, out[1][1] * 1e3), flush=True)
