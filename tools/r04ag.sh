#!/bin/bash
# direct-store epilogue of the all-DMA kernel (H2A_DIRECT=1): fuzz gate, then timings beside the staged epilogue
OUT=gpurun_out/r04ag; mkdir -p $OUT
GB=tools/gemm_bench
{ ok=1; for cfg in "5400 1" "5400 21"; do set -- $cfg; H2A_DIRECT=1 timeout 300 $GB fuzz $1 $2 16 31 | tail -2 | tee $OUT/fuzz_last.txt; grep -q "0 of 16 cases failed" $OUT/fuzz_last.txt || ok=0; done; } > $OUT/gate.txt 2>&1
cat $OUT/gate.txt
if [ $ok != 1 ]; then echo "GATE FAILED"; exit 0; fi
{ for M in 100 65 500; do for tn in 21 1; do for dr in 0 1; do echo "== M=$M tile $tn H2A_DIRECT=$dr"; H2A_DIRECT=$dr GEMM_PLAN_ALIGNED=4 timeout 120 $GB $M 256 4 5400 $tn | grep -E "^S[1256]|step GEMMs"; done; done; done; } 2>&1 | tee $OUT/h2a_direct_epilogue.txt
