#!/bin/bash
OUT=gpurun_out/r04f; mkdir -p $OUT
GB=tools/gemm_bench
{
echo "== fuzz (gate) of the LDS-DMA wide kernel"; ok=1
for cfg in "4 5250 1" "3 5250 1" "4 5250 21"; do set -- $cfg; H2D_NW=$1 timeout 300 $GB fuzz $2 $3 12 13 | tail -3 | tee $OUT/fuzz_last.txt; grep -q "0 of 12 cases failed" $OUT/fuzz_last.txt || ok=0; done
} > $OUT/gate.txt 2>&1
cat $OUT/gate.txt
if [ $ok != 1 ]; then echo "GATE FAILED: no timings"; timeout 120 $GB 500 256 4 5250 1 | head -5; exit 0; fi
{
for nw in 4 3; do echo "== h2d 128x256 NW=$nw"; for M in 500 2000; do H2D_NW=$nw timeout 120 $GB $M 256 4 5250 1 | grep -E "^S[1256]|step GEMMs|correctness|accuracy"; H2D_NW=$nw GEMM_PLAN_ALIGNED=4 timeout 120 $GB $M 256 4 5250 1 | grep -E "step GEMMs"; done; done
echo "== h2 (register-staged W) for comparison"; for M in 500 2000; do timeout 120 $GB $M 256 4 5200 1 | grep -E "^S[1256]|step GEMMs"; done
echo "== 128x128: h2d NW=4 vs h2, aligned 4"; for M in 500 100; do H2D_NW=4 GEMM_PLAN_ALIGNED=4 timeout 120 $GB $M 256 4 5250 21 | grep -E "^S[1256]|step GEMMs"; GEMM_PLAN_ALIGNED=4 timeout 120 $GB $M 256 4 5200 21 | grep -E "step GEMMs"; done
} > $OUT/h2d.txt 2>&1
cat $OUT/h2d.txt
