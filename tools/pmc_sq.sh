#!/bin/bash
# SQ counters of the kernels matching PATTERN in one run of a command   usage (via gpurun): bash tools/pmc_sq.sh PATTERN TAG -- python3 bench.py ...
set -e
PAT=$1; TAG=$2; shift; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/tmp_sq_$TAG -- "$@" > $O/sq_$TAG.out 2> $O/sq_$TAG.err || true
f=$(find $O/tmp_sq_$TAG -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/sq_$TAG.csv
rm -rf $O/tmp_sq_$TAG
python3 - "$PAT" <<PY
import csv,sys,collections
pat=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open("$O/sq_$TAG.csv")):
    k=r["Kernel_Name"].split("(")[0]
    if pat not in k: continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,d in acc.items():
    print(k[:80])
    for c,v in sorted(d.items()): print("   %-22s %.4g  (%d dispatches)"%(c,v,n[(k,c)]))
    wc=d.get("SQ_WAVE_CYCLES",0)
    if wc:
        for c in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_ANY"):
            if c in d: print("   %s / SQ_WAVE_CYCLES = %.3f"%(c,d[c]/wc))
PY
