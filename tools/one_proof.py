"""Profiling helper (not part of the product or the tests): proves `count` 2^logn square-chain statements one at a time with the
bench's tables installed, so that per-proof kernel statistics / PMC counters / latencies can be read off a run directly.
usage: python tools/one_proof.py [logn=20] [count=1] [--quad-max POINTS] [--table-rounds 1|2] [--curve 0|1] [--freeze-len N]"""
import argparse
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ark_bulletproofs_amd as A  # noqa: E402
from ark_bulletproofs_amd import engine as E  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("logn", nargs="?", type=int, default=20)
ap.add_argument("count", nargs="?", type=int, default=1)
ap.add_argument("--quad-max", type=int, default=0, help="BP_TUNE_FOLD_QUAD_MAX: fold rounds with at most this many points run four lanes per point")
ap.add_argument("--table-rounds", type=int, default=2, help="fold rounds served by the fixed-base tables (2: tables over 3N/4 bases)")
ap.add_argument("--curve", type=int, default=0)
ap.add_argument("--freeze-len", type=int, default=0, help="BP_TUNE_IPA_FREEZE_LEN: vector length from which G and H are no longer folded (0 = the library's default)")
args = ap.parse_args()
N = 1 << args.logn
eng = A.Engine(curve=args.curve)
eng.gens_derive(N)
eng.gens_fold_tables(N * 3 // 4 if args.table_rounds >= 2 else N // 2)
eng.gens_msm_tables(N)
if args.quad_max:
    eng.set_tuning(8, args.quad_max)
if args.freeze_len:
    eng.set_tuning(2, args.freeze_len)
eng.set_profiling(True)
lat = []
for k in range(args.count):
    st = E.Statement(args.curve, E.SC_SQUARE_CHAIN, [N, 0], bytes([9, k]) + bytes([3]) * 30)
    st.precompute()
    eng.reset_profiling()
    t0, c0 = time.perf_counter(), time.process_time()
    proof, tm = st.prove(eng)
    lat.append((time.perf_counter() - t0, tm[7], time.process_time() - c0))
    st.free()
names = {0: "msm accumulate", 9: "msm accumulate (fixed shape)", 10: "msm reduce + aggregate", 3: "fold (all)", 6: "fold tables", 7: "fold ladders", 8: "fold finish"}
kt = {k: eng.kernel_time(k) for k in names}
print("proved %d x 2^%d (curve %d, table rounds %d, quad max %d, freeze length %s), last proof %d bytes" % (args.count, args.logn, args.curve, args.table_rounds, args.quad_max, args.freeze_len or "default", len(proof)))
print("prove() wall after the TranscriptRng head: %s ms;  inner-product argument alone: %s ms;  host CPU time of the process in prove(): %s ms"
      % (", ".join("%.1f" % (a * 1e3) for a, _, _ in lat), ", ".join("%.1f" % (b * 1e3) for _, b, _ in lat), ", ".join("%.1f" % (c * 1e3) for _, _, c in lat)))
print("last proof, HIP-event times: " + "; ".join("%s %.2f ms / %d launches" % (names[k], kt[k][0], kt[k][1]) for k in names))
eng.close()
