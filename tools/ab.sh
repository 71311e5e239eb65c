#!/bin/bash
# One-call A/B on the GPU box: an optional parity gate, then bench.py values for a list of environment settings, repeated.
#   tools/ab.sh <tag> [-g "<pytest args>"] [-w "<workloads>"] [-r <repeats>] [-s <steps>] "<ENV=..;ENV=..>" ["<ENV=..>" ...]
# e.g. tools/ab.sh r05a -g "tests/test_gpu_headline.py" -w "beam5 greedy b13" "VSR_FUSE=1" "VSR_FUSE=0"
# Workloads: beam5 greedy xe scst b13 (= beam5 --batch 13) xe_real beam5_eval; b13nb / beam5nb / greedynb = the same with --rows-bound (no host sync in prepare).  Every configuration runs inside THIS call (boxes of the
# pool differ by up to 10 %: only same-call pairs are compared).  Replaces round 4's 22 single-use tools/r04*.sh (git log -- tools/).
TAG=$1; shift
GATE=""; WL="beam5 greedy b13"; REP=2; STEPS=30
while getopts "g:w:r:s:" o; do case $o in g) GATE=$OPTARG;; w) WL=$OPTARG;; r) REP=$OPTARG;; s) STEPS=$OPTARG;; esac; done
shift $((OPTIND - 1))
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -n "$GATE" ]; then
  timeout 1500 python -m pytest $GATE -m gpu -x -q 2>&1 | tail -15 > $OUT/gate.txt
  cat $OUT/gate.txt
  grep -q "passed" $OUT/gate.txt && ! grep -q "failed\|error" $OUT/gate.txt || echo "GATE FAILED (timings below are of a build that does not pass)"
fi
for rep in $(seq 1 $REP); do for cfg in "$@"; do for w in $WL; do
  case $w in b13) args="--workload beam5 --batch 13";; b13nb) args="--workload beam5 --batch 13 --rows-bound";; beam5nb) args="--workload beam5 --rows-bound";; greedynb) args="--workload greedy --rows-bound";; *) args="--workload $w";; esac
  env $(echo "$cfg" | tr ';' ' ') timeout 400 python bench.py $args --steps $STEPS --warmup 5 --no-cpu --no-secondary --no-alt 2>$OUT/last.err | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
    print('%-40s %-10s rep $rep %11.0f %-10s %8.3f ms  gemm avg %6.1f us share %.2f' % ('$cfg', '$w', d['value'], d['unit'], d['ms_per_step'], r.get('avg_launch_us', 0), r.get('gemm_share_of_wall', 0)))
except Exception as e:
    print('$cfg $w FAILED', e)
"
done; done; done 2>&1 | tee $OUT/ab.txt
