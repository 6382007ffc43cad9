#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box into gpurun_out/profiles_r03/ (copied to profiles/ afterwards).
# usage (via gpurun): bash tools/collect_profiles.sh
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {   # stats TAG -- cmd...
  tag=$1; shift; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$tag -- "$@" > $O/${tag}.out 2> $O/${tag}.err || true
  f=$(find $O/tmp_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r03_${tag}_kernel_stats.csv
  rm -rf $O/tmp_$tag
}
pmc() {     # pmc TAG -- cmd...
  tag=$1; shift; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/tmp_${tag}_$c -- "$@" > /dev/null 2> $O/${tag}_$c.err || true
    f=$(find $O/tmp_${tag}_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/${tag}_$c.csv
    rm -rf $O/tmp_${tag}_$c
  done
  python3 $R/tools/pmc_summary.py $tag $O/${tag}_FETCH_SIZE.csv $O/${tag}_WRITE_SIZE.csv $O/r03_pmc_${tag}_summary.json
  rm -f $O/${tag}_FETCH_SIZE.csv $O/${tag}_WRITE_SIZE.csv
}
stats prove_single -- python3 $R/tools/one_proof.py 20 2
stats bench_default -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline
stats verify -- python3 $R/bench.py --workload verify --steps 5 --warmup 1 --verify-inflight 1 --no-cpu-baseline   # (one batch at a time: kernel durations as bench.py reports them, from a batch run alone)
stats msm_2p16 -- python3 $R/bench.py --workload msm --terms 65536 --steps 20 --warmup 3 --no-cpu-baseline
stats msm_2p20 -- python3 $R/bench.py --workload msm --terms 1048576 --steps 10 --warmup 3 --no-cpu-baseline
stats shuffle_sweep -- python3 $R/bench.py --workload shuffle-sweep --steps 3 --warmup 1 --no-cpu-baseline --sweep-one-curve
pmc prove2p20 -- python3 $R/tools/one_proof.py 20 1
pmc verify4096 -- python3 $R/bench.py --workload verify --steps 1 --warmup 0 --no-cpu-baseline
pmc msm -- python3 $R/bench.py --workload msm --terms 65536 --steps 1 --warmup 0 --no-cpu-baseline
ls -la $O
