#!/bin/bash
# f32x3 kernel variants: fuzz + timings on the decoder shapes (M = 500 and M = 100), stream-K and k-aligned plans
OUT=gpurun_out/${1:-x3}; mkdir -p $OUT
for v in "3300 1" "3301 22" "3311 22" "3301 21" "3311 21" "3301 11" "3311 11"; do
  timeout 300 tools/gemm_bench fuzz $v 12 7 2>&1 | tail -1 | sed "s/^/fuzz $v: /"
done 2>&1 | tee $OUT/fuzz.txt
for M in 500 100; do
  for v in "1 1" "3300 1" "3301 22" "3311 22" "3311 21" "3311 11" "3301 11"; do
    for al in 0 8; do
      [ "$v" = "1 1" ] && [ $al = 8 ] && continue
      slots=256; [ "$v" = "1 1" ] && slots=768
      if [ $al = 0 ]; then r=$(timeout 120 tools/gemm_bench $M $slots 4 $v 2>&1 | grep -E "^S[0-9]|^step|correctness" | tr '\n' '|');
      else r=$(GEMM_PLAN_ALIGNED=$al timeout 120 tools/gemm_bench $M $slots 4 $v 2>&1 | grep -E "^S[0-9]|^step|correctness" | tr '\n' '|'); fi
      echo "M=$M v=$v aligned=$al: $r"
    done
  done
done 2>&1 | tee $OUT/timing.txt
