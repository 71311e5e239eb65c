#!/bin/bash
OUT=gpurun_out/r04d; mkdir -p $OUT
GB=tools/gemm_bench
{
echo "== fuzz (gate)"; ok=1
for v in "5200 1" "5200 21" "5300 1"; do timeout 300 $GB fuzz $v 10 11 | tail -1 | tee $OUT/fuzz_last.txt; grep -q "0 of 10 cases failed" $OUT/fuzz_last.txt || ok=0; done
} > $OUT/gate.txt 2>&1
cat $OUT/gate.txt
if [ $ok != 1 ]; then echo "GATE FAILED: stopping"; exit 0; fi
{
echo "== wide 128x256 (plane pad, packed exponents)"; for M in 500 2000; do timeout 120 $GB $M 256 4 5200 1 | grep -E "^S[1256]|step GEMMs|correctness"; GEMM_PLAN_ALIGNED=4 timeout 120 $GB $M 256 4 5200 1 | grep -E "step GEMMs"; done
echo "== 128x128 aligned 4"; for M in 500 100 13; do GEMM_PLAN_ALIGNED=4 timeout 120 $GB $M 256 4 5200 21 | grep -E "^S[1256]|step GEMMs"; done
echo "== stamps"; for M in 500 2000; do timeout 120 tools/gemm_bench_stamp $M 256 4 5200 1 | grep -E "^S[1256]|per k-tile|step GEMMs"; done
GEMM_PLAN_ALIGNED=4 timeout 120 tools/gemm_bench_stamp 500 256 4 5200 1 | grep -E "^S[1256]|per k-tile|step GEMMs"
} > $OUT/h2_wide.txt 2>&1
cat $OUT/h2_wide.txt
timeout 900 python -m pytest tests/test_gpu_attention_split.py tests/test_gpu_headline.py tests/test_gpu_h2.py -m gpu -q -x -s > $OUT/pytest_subset.log 2>&1; tail -12 $OUT/pytest_subset.log
bash tools/bench_matrix.sh r04d_att "VSR_ATT_SPLIT=1" "VSR_ATT_SPLIT=0"
