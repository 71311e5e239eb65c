#!/bin/bash
# all-DMA bf16 kernel (gemm_b16a.h, gemm_bench variant 1666): fuzz gate, bf16 tests, then VSR_B16_DMA = 1 / 0 on the bf16 workloads
OUT=gpurun_out/r04ab; mkdir -p $OUT
GB=tools/gemm_bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{ ok=1; for cfg in "1666 1" "1666 21"; do set -- $cfg; timeout 300 $GB fuzz $1 $2 12 13 | tail -2 | tee $OUT/fuzz_last.txt; grep -q "0 of 12 cases failed" $OUT/fuzz_last.txt || ok=0; done; } > $OUT/gate.txt 2>&1
cat $OUT/gate.txt
if [ $ok != 1 ]; then echo "GATE FAILED"; exit 0; fi
for M in 500 100; do for v in 1666 1665; do echo "== $v M=$M"; GEMM_PLAN_ALIGNED=4 timeout 120 $GB $M 256 4 $v 1 | grep -E "^S[1256]|step GEMMs|correctness"; done; done 2>&1 | tee $OUT/b16a_gemm_bench.txt
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/tests_bf16.txt
for rep in 1 2; do for dm in 1 0; do
  echo "== VSR_B16_DMA=$dm rep $rep"
  VSR_B16_DMA=$dm timeout 300 python bench.py --dtype bf16 --steps 30 --warmup 5 --no-cpu --no-secondary --no-alt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam5 bf16', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  VSR_B16_DMA=$dm timeout 300 python bench.py --dtype bf16 --workload greedy --steps 30 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('greedy bf16', d['value'], d['ms_per_step'])"
  VSR_B16_DMA=$dm timeout 300 python bench.py --dtype bf16 --workload xe --steps 20 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('xe bf16', d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $OUT/b16_dma_ab.txt
