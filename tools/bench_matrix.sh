#!/bin/bash
# bench.py values for a list of env settings: tools/bench_matrix.sh <tag> "<ENV=..;ENV=..>" ...   (each arg = one configuration)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for cfg in "$@"; do
  i=$((i+1))
  for w in beam5 greedy xe "beam5 --batch 13"; do
    name=$(echo "$w" | tr -d ' -')
    env $(echo $cfg | tr ';' ' ') timeout 300 python bench.py --workload $w --no-cpu --no-secondary --no-alt --steps 10 > $OUT/c${i}_$name.json 2> $OUT/c${i}_$name.err
    python3 -c "
import json,sys
try:
    d=json.load(open('$OUT/c${i}_$name.json')); r=d['roofline']
    print('%-28s %-16s %10.0f %s  %.3f ms  gemm %.1f TF/s avg %.1f us share %.2f'%('$cfg','$name',d['value'],d['unit'],d['ms_per_step'],r['achieved'],r['avg_launch_us'],r['gemm_share_of_wall']))
except Exception as e: print('$cfg $name FAILED', e)
"
  done
done 2>&1 | tee $OUT/summary.txt
