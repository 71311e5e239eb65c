#!/bin/bash
# ablations of the all-DMA wide kernel (tools/h2a_ablate.py): what paces a k-tile?
OUT=gpurun_out/r04s; mkdir -p $OUT
GB=tools/gemm_bench_abl
{
for M in 500 2000; do for abl in 0 1 2 3 4 5; do
  echo "== M=$M H2A_ABL=$abl"; GEMM_NOCHECK=1 GEMM_PLAN_ALIGNED=4 timeout 120 ${GB}$abl $M 256 4 5400 1 | grep -E "^S[15]|step GEMMs"
done; done
} 2>&1 | tee $OUT/h2a_ablations.txt
