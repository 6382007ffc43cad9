#!/bin/bash
# kernel statistics of the small-statement path: rocprofv3 --kernel-trace --stats over tools/exp_small_trace.py (k-shuffle proofs, one at a time)
# usage (via gpurun): bash tools/prof_small_r04.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_small; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for k in 2 128 1024; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k$k -- python3 $R/tools/exp_small_trace.py $k > $O/k$k.log 2>&1 || exit 1
  f=$(ls $O/k$k/*/*kernel_stats.csv | head -1); cp $f $O/r04_small_k${k}_kernel_stats.csv
  (cd $R && python tools/trace_tail.py gpurun_out/profiles_small/k$k 40 > $O/r04_small_k${k}_last_proof_trace.txt)
  tail -1 $O/k$k.log
done
