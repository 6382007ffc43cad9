#!/bin/bash
# Accumulate A/B, all arms in ONE session on one box:
#   (ref) lane per 16-entry chunk (k_msm_accum), (b) one wavefront per bucket (k_msm_accum_wave, ARKBP_MSM_ACCUM=wave),
#   (a) batched-affine additions through memory vs mixed adds (tools/ubench: k_baff vs k_madd_mem and the in-register madd rate).
# Writes gpurun_out/accum_ab/*.  usage: bash tools/exp_accum_ab.sh
set -e
out=gpurun_out/accum_ab; mkdir -p $out
for n in 65536 1048576; do
  python bench.py --workload msm --terms $n --steps 20 --warmup 3 --no-cpu-baseline > $out/chunk_$n.json 2> $out/chunk_$n.err
  ARKBP_MSM_ACCUM=wave python bench.py --workload msm --terms $n --steps 20 --warmup 3 --no-cpu-baseline > $out/wave_$n.json 2> $out/wave_$n.err
done
./tools/ubench > $out/ubench.txt
python - <<'PY'
import json
for n in (65536, 1048576):
    for arm in ("chunk", "wave"):
        d = json.load(open("gpurun_out/accum_ab/%s_%d.json" % (arm, n)))
        r = d["roofline"]
        print("%-6s n=%-8d  accumulate %.3f ms/launch  all MSM kernels %.3f ms  wall %.3f ms/MSM  %.1f M terms/s" % (arm, n, r["avg_kernel_ms"], r["msm_all_kernels_ms"], d["ms_per_step"], d["value"] / 1e6))
PY
grep -E "madd|batched|jac_madd secq   waves/SIMD=8" $out/ubench.txt
