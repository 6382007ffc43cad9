#!/bin/bash
# wall time per stand-alone MSM for 2^LO .. 2^HI terms, fixed-shape pipeline vs the general path (usage via gpurun: bash tools/sweep_sizes.sh 10 19)
for L in $(seq ${1:-10} ${2:-19}); do
  a=$(python3 $GRAFT_REPO_ROOT/bench.py --workload msm --terms $((1<<L)) --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
  b=$(ARKBP_MSM_NOFS=1 python3 $GRAFT_REPO_ROOT/bench.py --workload msm --terms $((1<<L)) --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
  echo "2^$L  fixed-shape $a ms   general $b ms"
done
