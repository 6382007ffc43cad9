"""Experiment helper (not part of the product): reads a rocprofv3 kernel trace CSV and reports how busy the GPU was —
the union of all kernel intervals over the span of the trace — plus the per-kernel share of the summed durations.
usage: python tools/gpu_busy.py <..._kernel_trace.csv> [t_skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
lo = t0 + (t1 - t0) * skip
iv = [(max(s, lo), e, k) for s, e, k in iv if e > lo]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - lo
per = defaultdict(int)
for s, e, k in iv:
    per[k.split("(")[0][:60]] += e - s
tot = sum(per.values())
print("span %.1f ms  busy(union) %.1f ms = %.1f %%   sum of durations %.1f ms (overlap factor %.2f)" % (span / 1e6, busy / 1e6, 100.0 * busy / span, tot / 1e6, tot / busy))
for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-62s %8.1f ms  %5.1f %%" % (k, v / 1e6, 100.0 * v / tot))

# idle gaps: where the union of kernel intervals is interrupted, and which kernels end / start around the largest ones
if len(sys.argv) > 3:
    gaps = []
    cur_e, last_k = None, None
    for s, e, k in iv:
        if cur_e is not None and s > cur_e:
            gaps.append((s - cur_e, last_k, k, cur_e))
        if cur_e is None or e > cur_e:
            cur_e, last_k = e, k
    gaps.sort(reverse=True)
    tot_gap = sum(g[0] for g in gaps)
    print("idle: %.1f ms in %d gaps; gaps > 50 us: %.1f ms in %d" % (tot_gap / 1e6, len(gaps), sum(g[0] for g in gaps if g[0] > 50000) / 1e6, sum(1 for g in gaps if g[0] > 50000)))
    by_next = defaultdict(int)
    by_prev = defaultdict(int)
    for g, pk, nk, _ in gaps:
        by_next[nk.split("(")[0][-44:]] += g
        by_prev[pk.split("(")[0][-44:]] += g
    print("idle time by the kernel that ENDS the gap (first to start):")
    for k, v in sorted(by_next.items(), key=lambda kv: -kv[1])[:8]:
        print("  %-46s %7.1f ms" % (k, v / 1e6))
    print("idle time by the kernel that ran last BEFORE the gap:")
    for k, v in sorted(by_prev.items(), key=lambda kv: -kv[1])[:8]:
        print("  %-46s %7.1f ms" % (k, v / 1e6))
