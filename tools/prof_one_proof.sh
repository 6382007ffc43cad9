#!/bin/bash
# kernel statistics of ONE 2^LOGN proof (tools/one_proof.py) -> gpurun_out/one_proof_TAG.csv + a per-kernel summary  (usage via gpurun: bash tools/prof_one_proof.sh 20 tag)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; L=${1:-20}; TAG=${2:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_op_$TAG -- python3 $R/tools/one_proof.py $L 2 > $O/one_proof_$TAG.out 2> $O/one_proof_$TAG.err || true
f=$(find $O/tmp_op_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/one_proof_$TAG.csv
rm -rf $O/tmp_op_$TAG
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/one_proof_$TAG.csv")))
tot=0
for r in rows[:28]:
    nm=r['Name'].split('(')[0].split('<')[0].split('::')[-1][:36]; print(f"{nm:38s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:8.2f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us")
print("all kernels: %.1f ms" % (sum(float(r['TotalDurationNs']) for r in rows)/1e6))
PY
tail -3 $O/one_proof_$TAG.out
