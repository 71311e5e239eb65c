#!/bin/bash
OUT=gpurun_out/r04af; mkdir -p $OUT
{ for M in 500 2000 100; do for abl in 0 6; do echo "== M=$M H2A_ABL=$abl"; GEMM_NOCHECK=1 GEMM_PLAN_ALIGNED=4 timeout 120 tools/gemm_bench_abl$abl $M 256 4 5400 1 | grep -E "^S[1256]|step GEMMs"; done; done; } 2>&1 | tee $OUT/h2a_epilogue_ablation.txt
