"""Profiling helper (not part of the product or the tests): proves exactly ONE 2^logn square-chain statement with the bench's
tables installed, so that per-proof kernel statistics / PMC counters can be read off a rocprofv3 run directly.
usage: python tools/one_proof.py [logn=20] [count=1]"""
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ark_bulletproofs_amd as A  # noqa: E402
from ark_bulletproofs_amd import engine as E  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = 1 << logn
eng = A.Engine(curve=0)
eng.gens_derive(N)
eng.gens_fold_tables(N // 2)
eng.gens_msm_tables(N)
for k in range(count):
    st = E.Statement(0, E.SC_SQUARE_CHAIN, [N, 0], bytes([9, k]) + bytes([3]) * 30)
    st.precompute()
    proof, tm = st.prove(eng)
    st.free()
print("proved %d x 2^%d, last proof %d bytes, prove() %.3f s" % (count, logn, len(proof), tm[0]))
eng.close()
