#!/bin/bash
# f16x2 kernels (csrc/gemm_h2.h) in tools/gemm_bench: fuzz, accuracy (incl. loosened bounds: fp16 subnormal behaviour), timings next to f32x3.  usage: tools/h2_round.sh <tag>
TAG=${1:-h2a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
GB=tools/gemm_bench
{
echo "== fuzz"; if false; then
for v in "5200 1" "5200 21" "5300 1"; do timeout 300 $GB fuzz $v 24 3 | tail -4; done
H2S_NS=2 timeout 300 $GB fuzz 5300 1 24 5 | tail -2; fi
echo "== accuracy, tight and loosened bounds (H2_LOOSE bits)"
for l in 0 4 8 12; do echo "-- H2_LOOSE=$l"; H2_LOOSE=$l timeout 120 $GB 500 256 4 5200 1 | grep -E "accuracy|correctness"; done
echo "-- f32x3 for comparison"; timeout 120 $GB 500 256 4 3300 1 | grep -E "accuracy|correctness"
echo "== M=500: f16x2 wide vs f32x3 wide (stream-K, then k-aligned 4)"
for v in "5200 1" "3300 1"; do timeout 120 $GB 500 256 4 $v | grep -E "^S[1256]|step GEMMs"; GEMM_PLAN_ALIGNED=4 timeout 120 $GB 500 256 4 $v | grep -E "^S[1256]|step GEMMs"; done
echo "== M=500, 128x128 tiles"
for v in "5200 21" "3300 21"; do GEMM_PLAN_ALIGNED=4 timeout 120 $GB 500 256 4 $v | grep -E "^S[1256]|step GEMMs"; done
echo "== M=2000 (training batched GEMMs)"
for v in "5200 1" "3300 1"; do GEMM_PLAN_ALIGNED=4 timeout 120 $GB 2000 256 4 $v | grep -E "^S[1256]|step GEMMs"; done
for M in 100 65 32 13; do
  echo "== M=$M: streaming f16x2 (NS 1 / 2, slots 512 / 768 / 256) vs f16x2 128x128 tile vs f32x3 128x128 tile vs f32x3 streaming"
  for ns in 1 2; do for sl in 512 768 256; do for mi in 8 4; do echo "-- h2s NS=$ns slots=$sl min=$mi"; H2S_NS=$ns GEMM_PLAN_ALIGNED=$mi timeout 120 $GB $M $sl 4 5300 0 | grep -E "^S[1256]|step GEMMs"; done; done; done
  echo "-- h2 128x128 aligned 4"; GEMM_PLAN_ALIGNED=4 timeout 120 $GB $M 256 4 5200 21 | grep -E "^S[1256]|step GEMMs"
  echo "-- x3 128x128 aligned 4"; GEMM_PLAN_ALIGNED=4 timeout 120 $GB $M 256 4 3300 21 | grep -E "^S[1256]|step GEMMs"
  echo "-- x3s NS=1 slots 512"; X3S_NS=1 GEMM_PLAN_ALIGNED=8 timeout 120 $GB $M 512 4 3400 0 | grep -E "^S[1256]|step GEMMs"
done
} > $OUT/h2_gemm_bench.txt 2>&1
tail -150 $OUT/h2_gemm_bench.txt
