#!/bin/bash
OUT=gpurun_out/r04c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_attention_split.py tests/test_gpu_headline.py tests/test_gpu_h2.py -m gpu -q -x -s > $OUT/pytest_subset.log 2>&1; tail -15 $OUT/pytest_subset.log
bash tools/bench_matrix.sh r04c_att "VSR_ATT_SPLIT=1" "VSR_ATT_SPLIT=0"
GB=tools/gemm_bench
{
for pd in 2 3; do echo "== wide 128x256 H2_PD=$pd"; for M in 500 2000; do H2_PD=$pd timeout 120 $GB $M 256 4 5200 1 | grep -E "^S[1256]|step GEMMs|correctness"; done; done
for pd in 2 3 4; do echo "== 128x128 H2_PD=$pd aligned 4"; for M in 500 2000; do H2_PD=$pd GEMM_PLAN_ALIGNED=4 timeout 120 $GB $M 256 4 5200 21 | grep -E "^S[1256]|step GEMMs|correctness"; done; done
echo "== fuzz of the PD variants"; for pd in 3; do H2_PD=$pd timeout 300 $GB fuzz 5200 1 12 7 | tail -1; done; for pd in 3 4; do H2_PD=$pd timeout 300 $GB fuzz 5200 21 12 7 | tail -1; done
} > $OUT/h2_pd.txt 2>&1
cat $OUT/h2_pd.txt
for v in 1 0; do echo "== VSR_BF16_P_FP32=$v"; VSR_BF16_P_FP32=$v timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q -s -k "batch100 or wide" 2>&1 | grep -E "bf16 vs fp32|passed|failed|max \|" ; done > $OUT/bf16_p.txt 2>&1
cat $OUT/bf16_p.txt
