"""Experiment driver (not part of the product or the tests): cfg3's shuffle statement at full size.
usage: python tools/exp_shuffle.py LOGK   (k = 2^LOGK + 1 inputs -> 2^(LOGK+1) multipliers, m = 2k commitments)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ark_bulletproofs_amd as A  # noqa: E402
from ark_bulletproofs_amd import engine as E  # noqa: E402

logk = int(sys.argv[1])
k = (1 << logk) + 1
N = 2 * (k - 1)
eng = A.Engine(curve=0)
t0 = time.perf_counter()
eng.gens_derive(N)
print("gens %.2f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
st = E.Statement(0, E.SC_SHUFFLE, [k], bytes([3]) * 32, engine=eng)
t_stmt = time.perf_counter() - t0
commits, pubs, nm, nq = st.info(m_cap=2 * k + 8)
print("statement: k=%d m=%d multipliers=%d constraints=%d  built in %.2f s (commits on the GPU)" % (k, len(commits), nm, nq, t_stmt), flush=True)
t0 = time.perf_counter()
proof, tm = st.prove(eng)
t_prove = time.perf_counter() - t0
print("prove %.2f s  stages %s  proof %d B" % (t_prove, ["%.3f" % x for x in tm], len(proof)), flush=True)
t0 = time.perf_counter()
rc = eng.verify_scenario(E.SC_SHUFFLE, [k], proof, commits, pubs)
print("verify rc=%d in %.2f s" % (rc, time.perf_counter() - t0), flush=True)
bad = bytearray(proof)
bad[40] ^= 1
print("tampered rc=%d" % eng.verify_scenario(E.SC_SHUFFLE, [k], bytes(bad), commits, pubs))
