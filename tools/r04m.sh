#!/bin/bash
# all-DMA kernel in the product: gate (fuzz + f16x2 tests + fixtures), then beam-5 / greedy / batch-13 / xe beside VSR_H2_AIMG=0
OUT=gpurun_out/r04m; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_gemm_fuzz.py tests/test_gpu_h2.py tests/test_gpu_parity.py tests/test_gpu_headline.py -m gpu -x -q 2>&1 | tail -15 > $OUT/tests.txt
cat $OUT/tests.txt
grep -q "passed" $OUT/tests.txt && ! grep -q "failed\|error" $OUT/tests.txt || { echo "GATE FAILED"; exit 0; }
for rep in 1 2; do for ai in 1 0; do
  echo "== VSR_H2_AIMG=$ai rep $rep"
  VSR_H2_AIMG=$ai timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu --no-secondary --no-alt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam5', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  VSR_H2_AIMG=$ai timeout 300 python bench.py --workload greedy --steps 30 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('greedy', d['value'], d['ms_per_step'])"
  VSR_H2_AIMG=$ai timeout 300 python bench.py --batch 13 --steps 30 --warmup 5 --no-cpu --no-secondary --no-alt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch13', d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $OUT/aimg_ab.txt
