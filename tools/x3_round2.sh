#!/bin/bash
OUT=gpurun_out/${1:-x3b}; mkdir -p $OUT
for M in 100 65 13; do
  for v in "1 1" "1607 2" "3300 21" "3301 21" "3311 21" "3300 11" "3301 11"; do
    for al in 0 8 4; do
      case "$v" in "1 1"|"1607 2") [ $al != 0 ] && continue;; esac
      slots=256; [ "$v" = "1 1" ] && slots=768
      [ "$v" = "1607 2" ] && [ $M != 100 ] && v="160$(( (M+15)/16 )) 2"
      if [ $al = 0 ]; then r=$(timeout 120 tools/gemm_bench $M $slots 4 $v 2>&1 | grep -E "^S[0-9]|^step" | tr '\n' '|');
      else r=$(GEMM_PLAN_ALIGNED=$al timeout 120 tools/gemm_bench $M $slots 4 $v 2>&1 | grep -E "^S[0-9]|^step" | tr '\n' '|'); fi
      echo "M=$M v=$v aligned=$al: $r"
    done
  done
done 2>&1 | tee $OUT/timing.txt
