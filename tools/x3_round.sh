#!/bin/bash
# f32x3 kernel variants on the decoder's step shapes: fuzz + timings, stream-K ranges and k-aligned pieces.  usage: tools/x3_round.sh <tag>
OUT=gpurun_out/${1:-x3}; mkdir -p $OUT
for v in "3300 1" "3300 21" "3400 1" "1664 21" "1665 21"; do
  timeout 300 tools/gemm_bench fuzz $v 12 7 2>&1 | tail -1 | sed "s/^/fuzz $v: /"
done 2>&1 | tee $OUT/fuzz.txt
run() {   # M variant tn aligned slots
  if [ $4 = 0 ]; then r=$(timeout 120 tools/gemm_bench $1 $5 4 $2 $3 2>&1 | grep -E "^S[0-9]|^step" | tr '\n' '|');
  else r=$(GEMM_PLAN_ALIGNED=$4 timeout 120 tools/gemm_bench $1 $5 4 $2 $3 2>&1 | grep -E "^S[0-9]|^step" | tr '\n' '|'); fi
  echo "M=$1 v=$2 $3 aligned=$4 slots=$5: $r" | sed "s/nslab //g; s/ TF\/s//g"
}
{
  for al in 0 4; do run 500 3300 1 $al 256; run 500 3300 21 $al 256; done
  run 500 1 1 0 768
  for M in 100 65 32 13; do
    run $M 3300 21 4 256
    X3S_NS=1 run $M 3400 1 8 512
    X3S_NS=2 run $M 3400 1 8 256
    run $M 1 1 0 768
    run $M 160$(( (M > 112 ? 112 : M + 15) / 16 )) 2 0 256
  done
} 2>&1 | tee $OUT/timing.txt
