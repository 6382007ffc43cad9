import sys, time, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ark_bulletproofs_amd as A
n = 1 << 22
eng = A.Engine(curve=0)
eng.gens_derive(n // 2)
rng = np.random.default_rng(1)
sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); sc[:, 3] >>= np.uint64(2)
for _ in range(2): eng.msm_gens(n // 2, sc)
t0, c0 = time.perf_counter(), time.process_time()
for _ in range(10): eng.msm_gens(n // 2, sc)
w, c = time.perf_counter() - t0, time.process_time() - c0
print("10 x 2^22-term MSM over the resident generators: wall %.1f ms, process CPU %.1f ms (%.2f cores)" % (w * 1e3, c * 1e3, c / w))
