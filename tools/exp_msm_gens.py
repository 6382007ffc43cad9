"""Experiment driver (not part of the product or the tests): an MSM over the resident generator tables G[0..n) || H[0..n) with random
scalars, through the ordinary schedule and through the fixed-base rows (bp_gens_msm_tables).  usage: python tools/exp_msm_gens.py LOGN"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ark_bulletproofs_amd as A  # noqa: E402

logn = int(sys.argv[1])
n = 1 << logn
eng = A.Engine(curve=0)
eng.gens_derive(n)
rng = np.random.default_rng(5)
sc = rng.integers(0, 1 << 63, size=(2 * n, 4), dtype=np.uint64)
sc[:, 3] >>= np.uint64(2)
for tables in (False, True):
    if tables:
        eng.gens_msm_tables(n)
    ref = None
    for _ in range(2):
        out = eng.msm_gens(n, sc)
    eng.set_profiling(True)
    eng.reset_profiling()
    t0 = time.perf_counter()
    for _ in range(5):
        out = eng.msm_gens(n, sc)
    dt = (time.perf_counter() - t0) / 5
    print("2 x 2^%d generator terms, %s: wall %.3f ms (with the 2^%d x 32 B scalar upload)  accumulate %.3f ms  all kernels %.3f ms  result %s"
          % (logn, "fixed-base rows" if tables else "ordinary schedule", dt * 1e3, logn + 1, eng.kernel_time(0)[0] / 5, eng.kernel_time(1)[0] / 5, hex(int(out[0]))[:12]))
eng.close()
