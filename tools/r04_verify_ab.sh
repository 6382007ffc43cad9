#!/bin/bash
# A/B of the batch verifier in ONE session (boxes differ by several percent): device front end vs host replay, batches in flight.
mkdir -p gpurun_out
for nf in 2 4 6; do
  python bench.py --workload verify --steps 24 --warmup 3 --verify-inflight $nf --no-cpu-baseline > gpurun_out/r04_vfy_dev_nf$nf.json 2> gpurun_out/r04_vfy_dev_nf$nf.err || exit 1
done
ARKBP_VFY_HOST=1 python bench.py --workload verify --steps 24 --warmup 3 --verify-inflight 4 --no-cpu-baseline > gpurun_out/r04_vfy_host_nf4.json 2> gpurun_out/r04_vfy_host_nf4.err || exit 1
python - <<'PY'
import json
for f in ["dev_nf2", "dev_nf4", "dev_nf6", "host_nf4"]:
    r = json.load(open("gpurun_out/r04_vfy_%s.json" % f))
    c = r["config"]
    print(f, "%.0f proofs/s" % r["value"], "%.2f ms/batch" % r["ms_per_step"], "cores %.1f" % c["host_cpu_in_timed_region"].get("avg_cores_used", -1), c["stage_ms_per_step"])
PY
