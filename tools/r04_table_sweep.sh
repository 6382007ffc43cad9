#!/bin/bash
# Two-round fold tables: prove throughput against the window width (HBM spent), ONE session.
mkdir -p gpurun_out
for w in 8 4 5 6 7 8; do
  python bench.py --workload prove --steps 16 --warmup 4 --fold-table-bits $w --tables-off-steps 0 --no-cpu-baseline > gpurun_out/r04_tabsweep_w$w.json 2> gpurun_out/r04_tabsweep_w$w.err || exit 1
  python - <<PY
import json
r = json.load(open("gpurun_out/r04_tabsweep_w$w.json"))
t = r["config"]["first_round_fold_tables"]; m = r["config"]["fixed_base_msm_tables"]
print("w=%d  %.2f M constraints/s  %.1f ms/step  tables %.1f GB (+%.1f GB msm rows)  build %.1f s" % (t["window_bits"], r["value"] / 1e6, r["ms_per_step"], t["GB"], m["GB"], t["build_s"]), flush=True)
PY
done
