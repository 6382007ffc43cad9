"""Experiment driver (not part of the product or the tests): times one MSM shape on the GPU for front-end tuning.
usage: python tools/exp_msm.py LOGN MODE   (MODE: full | low240 | mont)"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ark_bulletproofs_amd as A  # noqa: E402
from bench import synth_msm_inputs  # noqa: E402

logn, mode = int(sys.argv[1]), sys.argv[2]
n = 1 << logn
eng = A.Engine(curve=0)
bases, sc = synth_msm_inputs(eng, min(n, 1 << 14), 0)
reps = n // len(bases)
bases = np.tile(bases, (reps, 1))
rng = np.random.default_rng(5)
sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
canonical = mode != "mont"
if mode == "low240":
    sc[:, 3] >>= np.uint64(24)
else:
    sc[:, 3] >>= np.uint64(2)
db = eng.upload_points(bases)
ds = eng.upload_scalars(sc) if not canonical else eng.upload_scalars(sc)
for _ in range(2):
    eng.msm_dev(db, ds, n, canonical=canonical)
eng.set_profiling(True)
eng.reset_profiling()
t0 = time.perf_counter()
for _ in range(5):
    eng.msm_dev(db, ds, n, canonical=canonical)
dt = (time.perf_counter() - t0) / 5
print("n=2^%d mode=%s wall %.3f ms  accum %.3f ms  all kernels %.3f ms" % (logn, mode, dt * 1e3, eng.kernel_time(0)[0] / 5, eng.kernel_time(1)[0] / 5))
