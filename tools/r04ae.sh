#!/bin/bash
# streaming f16x2 kernel: prefetch depth (k-tiles of weights in flight per wave) 2 / 3 (shipped) / 4 / 5 / 6 at M = 65 and 13, slots 512 / 768
OUT=gpurun_out/r04ae; mkdir -p $OUT
{
for M in 65 13; do for pf in 2 3 4 5 6; do
  GB=tools/gemm_bench_pf$pf; [ $pf = 3 ] && GB=tools/gemm_bench
  for sl in 512 768; do echo "== M=$M PF=$pf slots=$sl"; GEMM_PLAN_ALIGNED=8 timeout 120 $GB $M $sl 4 5300 1 | grep -E "^S[1256]|step GEMMs|correctness"; done
done; done
} 2>&1 | tee $OUT/h2s_prefetch_depth.txt
