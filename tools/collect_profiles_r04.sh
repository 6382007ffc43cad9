#!/bin/bash
# Round-4 rocprofv3 evidence, collected on the GPU box into gpurun_out/profiles_r04/ (copied to profiles/ afterwards).
# usage (via gpurun): bash tools/collect_profiles_r04.sh [what ...]   what: calib stats pmc (default: all)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WHAT=${@:-calib stats pmc}
stats() { tag=$1; shift; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$tag -- "$@" > $O/${tag}.out 2> $O/${tag}.err || true
  f=$(find $O/tmp_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r04_${tag}_kernel_stats.csv; rm -rf $O/tmp_$tag; }
pmc() { tag=$1; shift; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/tmp_${tag}_$c -- "$@" > /dev/null 2> $O/${tag}_$c.err || true
    f=$(find $O/tmp_${tag}_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/${tag}_$c.csv; rm -rf $O/tmp_${tag}_$c
  done
  python3 $R/tools/pmc_summary.py $tag $O/${tag}_FETCH_SIZE.csv $O/${tag}_WRITE_SIZE.csv $O/r04_pmc_${tag}_summary.json
  rm -f $O/${tag}_FETCH_SIZE.csv $O/${tag}_WRITE_SIZE.csv; }
for w in $WHAT; do case $w in
calib)
  $R/tools/ubench_fetch > $O/r04_fetch_calibration.txt 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/tmp_cal -- $R/tools/ubench_fetch > /dev/null 2> $O/cal.err || true
  f=$(find $O/tmp_cal -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $O/r04_fetch_calibration.txt <<'PY'
import csv, sys
must = {"k_stream16": 4096.0, "k_gather<4>": 4096.0, "k_gather<2>": 4096.0}
print("rocprofv3 --pmc FETCH_SIZE of the same binary (KiB as reported -> MiB), against the MiB that must come from HBM (64-byte sectors):")
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]; k = "k_stream16" if "stream16" in n else "k_gather<4>" if "<4>" in n else "k_gather<2>" if "<2>" in n else None
    if k: print("  %-12s FETCH_SIZE %.1f MiB   must fetch %.1f MiB   counter / bytes = %.3f" % (k, float(r["Counter_Value"]) / 1024.0, must[k], float(r["Counter_Value"]) / 1024.0 / must[k]))
PY
  rm -rf $O/tmp_cal ;;
stats)
  stats prove_single -- python3 $R/tools/one_proof.py 20 2
  stats verify -- python3 $R/bench.py --workload verify --steps 5 --warmup 1 --verify-inflight 1 --no-cpu-baseline
  stats msm_2p16 -- python3 $R/bench.py --workload msm --terms 65536 --steps 20 --warmup 3 --no-cpu-baseline
  stats bench_default -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only ;;
pmc)
  pmc prove2p20 -- python3 $R/tools/one_proof.py 20 1
  pmc verify4096 -- python3 $R/bench.py --workload verify --steps 1 --warmup 0 --verify-inflight 1 --no-cpu-baseline
  pmc msm -- python3 $R/bench.py --workload msm --terms 65536 --steps 1 --warmup 0 --no-cpu-baseline ;;
esac; done
ls -la $O
