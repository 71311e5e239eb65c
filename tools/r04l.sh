#!/bin/bash
# all-DMA wide f16x2 kernel (gemm_h2a.h, variant 5400): fuzz gate, then timings beside the shipped wide kernel (5200)
OUT=gpurun_out/r04l; mkdir -p $OUT
GB=tools/gemm_bench
{
echo "== fuzz (gate)"; ok=1
for cfg in "5400 1" "5400 21"; do set -- $cfg; timeout 300 $GB fuzz $1 $2 12 13 | tail -3 | tee $OUT/fuzz_last.txt; grep -q "0 of 12 cases failed" $OUT/fuzz_last.txt || ok=0; done
} > $OUT/gate.txt 2>&1
cat $OUT/gate.txt
if [ $ok != 1 ]; then echo "GATE FAILED"; timeout 120 $GB 500 256 4 5400 1 | head -12; exit 0; fi
{
for M in 500 2000; do for v in 5400 5200; do echo "== $v 128x256 M=$M"; timeout 120 $GB $M 256 4 $v 1 | grep -E "^S[1256]|step GEMMs|correctness|accuracy"; GEMM_PLAN_ALIGNED=4 timeout 120 $GB $M 256 4 $v 1 | grep -E "step GEMMs"; done; done
for M in 500 100; do for v in 5400 5200; do echo "== $v 128x128 aligned 4 M=$M"; GEMM_PLAN_ALIGNED=4 timeout 120 $GB $M 256 4 $v 21 | grep -E "^S[1256]|step GEMMs"; done; done
} > $OUT/h2a.txt 2>&1
cat $OUT/h2a.txt
