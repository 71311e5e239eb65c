#!/bin/bash
OUT=gpurun_out/r04e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_bf16.py tests/test_gpu_headline.py tests/test_gpu_train.py -m gpu -q -x -s > $OUT/pytest_subset.log 2>&1; tail -12 $OUT/pytest_subset.log
grep -E "bf16 losses|bf16 vs fp32" $OUT/pytest_subset.log
bash tools/bench_matrix.sh r04e_route "VSR_H2S_MAX=128" "VSR_H2S_MAX=80" "VSR_H2S_MAX=48"
