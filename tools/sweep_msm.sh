#!/bin/bash
# sweep of (window bits, chunk size) for the fixed-shape MSM at 2^LOGN terms (usage via gpurun: bash tools/sweep_msm.sh 16 "11 12 13" "8 12 16"; the third list = entries per chunk)
L=${1:-16}
for c in $2; do for h in $3; do
  echo "== c=$c chunk=$h"; ARKBP_MSM_C=$c ARKBP_MSM_FS_CAP=$h bash $GRAFT_REPO_ROOT/tools/prof_msm.sh $L c${c}h${h} || exit 1
done; done
