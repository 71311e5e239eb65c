#!/bin/bash
# attention scores through tanh_fast (VSR_FAST_TANH=1) vs ocml tanhf: beam-5 / greedy, kernel stats
OUT=gpurun_out/r04p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for ft in 0 1; do
  echo "== VSR_FAST_TANH=$ft rep $rep"
  VSR_FAST_TANH=$ft timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu --no-secondary --no-alt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam5', d['value'], d['ms_per_step'])"
  VSR_FAST_TANH=$ft timeout 300 python bench.py --workload greedy --steps 30 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('greedy', d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $OUT/fast_tanh_ab.txt
VSR_FAST_TANH=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py tests/test_gpu_h2.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/tests_fast_tanh.txt
export VSR_FAST_TANH=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b5 -- python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-alt > $OUT/prof.log 2>&1
f=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-150 > $OUT/beam5_kernel_stats_fast_tanh.csv; cat $OUT/beam5_kernel_stats_fast_tanh.csv
