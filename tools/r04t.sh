#!/bin/bash
OUT=gpurun_out/r04t; mkdir -p $OUT
for rep in 1 2; do for ef in 75 80 70; do
  echo "== VSR_ALIGNED_EFF=$ef rep $rep"
  VSR_ALIGNED_EFF=$ef timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu --no-secondary --no-alt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam5', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done; done 2>&1 | tee $OUT/aligned_eff_ab.txt
