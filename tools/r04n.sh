#!/bin/bash
# small launches: the all-DMA 128 x 128 tile against the streaming kernel (VSR_H2S_MAX = rows up to which the streaming kernel is used)
OUT=gpurun_out/r04n; mkdir -p $OUT
for rep in 1 2; do for mx in 80 0 32; do
  echo "== VSR_H2S_MAX=$mx rep $rep"
  VSR_H2S_MAX=$mx timeout 300 python bench.py --batch 13 --steps 30 --warmup 5 --no-cpu --no-secondary --no-alt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch13 beam5', d['value'], d['ms_per_step'])"
  VSR_H2S_MAX=$mx timeout 300 python bench.py --workload greedy --batch 13 --steps 30 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch13 greedy', d['value'], d['ms_per_step'])"
  VSR_H2S_MAX=$mx timeout 300 python bench.py --workload greedy --batch 50 --steps 30 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch50 greedy', d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $OUT/small_ab.txt
