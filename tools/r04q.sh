#!/bin/bash
# wide launches with the all-DMA kernel: k-aligned pieces when >= 75 % efficient (21, default) / never (01) / always (11)
OUT=gpurun_out/r04q; mkdir -p $OUT
for rep in 1 2; do for al in 21 01 11; do
  echo "== VSR_X3_ALIGNED=$al rep $rep"
  VSR_X3_ALIGNED=$al timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu --no-secondary --no-alt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam5', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done; done 2>&1 | tee $OUT/aligned_ab.txt
