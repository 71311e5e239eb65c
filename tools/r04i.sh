#!/bin/bash
# backward pass on the f16x2 kernels: gradient tests first (gate), then the XE step timed beside the f32x3 backward
OUT=gpurun_out/r04w; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_indexed.py tests/test_gpu_h2.py tests/test_gpu_ssp.py -m gpu -x -q 2>&1 | tail -15 > $OUT/tests.txt
cat $OUT/tests.txt
grep -q "passed" $OUT/tests.txt && ! grep -q "failed" $OUT/tests.txt || { echo "GATE FAILED"; exit 0; }
for i in 1 2; do
  timeout 600 python bench.py --workload xe --steps 20 --warmup 5 --no-cpu 2>/dev/null | tail -1 > $OUT/xe_h2bwd_$i.json; cat $OUT/xe_h2bwd_$i.json | cut -c1-400
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o xe -- python bench.py --workload xe --steps 10 --warmup 3 --no-cpu > $OUT/prof.log 2>&1
f=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200 > $OUT/xe_kernel_stats.csv; cat $OUT/xe_kernel_stats.csv
