#!/bin/bash
# four-column LSTM pointwise kernels: parity gate, then beam-5 / greedy / batch-13 and the kernel stats
OUT=gpurun_out/r04r; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py tests/test_gpu_h2.py tests/test_gpu_configs.py tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -6 > $OUT/tests.txt
cat $OUT/tests.txt
grep -q "passed" $OUT/tests.txt && ! grep -q "failed\|error" $OUT/tests.txt || { echo "GATE FAILED"; exit 0; }
for rep in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu --no-secondary --no-alt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam5', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --workload greedy --steps 30 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('greedy', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --batch 13 --steps 30 --warmup 5 --no-cpu --no-secondary --no-alt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch13', d['value'], d['ms_per_step'])"
done 2>&1 | tee $OUT/bench.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b5 -- python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-alt > $OUT/prof.log 2>&1
f=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 "$f" > $OUT/beam5_kernel_stats.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/beam5_kernel_stats.csv")):
    print("%-50s %6s %9.1f us  %6s%%"%(r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
