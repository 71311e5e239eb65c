#!/bin/bash
# fewer, longer k-pieces (fewer slabs for the consumers; at the power limit idle CUs are traded for clock): VSR_H2_ALIGNED_MIN = min k-tiles per piece
OUT=gpurun_out/r04aa; mkdir -p $OUT
for rep in 1 2; do for mi in 4 24 40 60; do
  echo "== VSR_H2_ALIGNED_MIN=$mi rep $rep"
  VSR_H2_ALIGNED_MIN=$mi timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu --no-secondary --no-alt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam5', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done; done 2>&1 | tee $OUT/aligned_min_ab.txt
