#!/bin/bash
# kernel statistics of the stand-alone MSM bench at 2^LOGN terms -> gpurun_out/msm_prof_LOGN.csv   (usage via gpurun: bash tools/prof_msm.sh 16 [env...])
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; L=${1:-16}; TAG=${2:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_msm_$L -- python3 $R/bench.py --workload msm --terms $((1<<L)) --steps 20 --warmup 3 --no-cpu-baseline > $O/msm_prof_${L}_$TAG.out 2> $O/msm_prof_${L}_$TAG.err || true
f=$(find $O/tmp_msm_$L -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/msm_prof_${L}_$TAG.csv
rm -rf $O/tmp_msm_$L
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/msm_prof_${L}_$TAG.csv")))
tot=0
for r in rows:
    if 'k_msm' in r['Name']:
        nm=r['Name'].split('(')[0].split('<')[0].split('::')[-1][:40]; print(f"{nm:42s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us"); tot+=float(r['AverageNs'])*int(r['Calls'])/23
print("sum per msm %.1f us"%(tot/1e3))
PY
tail -1 $O/msm_prof_${L}_$TAG.out | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
