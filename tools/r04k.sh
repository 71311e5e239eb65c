#!/bin/bash
# beam-5 step: the vocabulary + next-LSTM1 launch (256 tiles, unequal K) on k-aligned pieces (W fetched once per XCD) vs stream-K ranges
OUT=gpurun_out/r04k; mkdir -p $OUT
for rep in 1 2; do for al in 21 11; do
  echo "== VSR_X3_ALIGNED=$al rep $rep"
  VSR_X3_ALIGNED=$al timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu --no-secondary --no-alt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  VSR_X3_ALIGNED=$al timeout 300 python bench.py --workload greedy --steps 30 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('greedy', d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $OUT/aligned_ab.txt
