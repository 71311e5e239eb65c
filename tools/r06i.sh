#!/bin/bash
# round 6, review item 5 (CU-partitioned overlap of two half-batches): what would the GEMM partition have to carry?
# Kernel stats of a 50-image (M = 250) beam-5 call whose GEMM plans are made for 200 workgroups (= a 200-CU partition), beside the whole batch.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06i; mkdir -p $OUT
for cfg in "50 200" "50 256" "100 256"; do set -- $cfg
  (cd /tmp && VSR_GEMM_SLOTS_BF16=$2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b$1_s$2 -- python3 $GRAFT_REPO_ROOT/bench.py --batch $1 --steps 10 --warmup 3 --no-cpu --no-secondary --no-alt > $OUT/stats_b$1_s$2.log 2>&1)
  tail -1 $OUT/stats_b$1_s$2.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $1 slots $2: %.0f tokens/s %.3f ms per call'%(d['value'], d['ms_per_step']))"
  f=$(find $OUT/stats_b$1_s$2 -name "*kernel_stats.csv"); python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
calls=13
g=sum(float(r['TotalDurationNs']) for r in rows if 'gemm_nt' in r['Name'])/calls/1e3
p=sum(float(r['TotalDurationNs']) for r in rows if any(k in r['Name'] for k in ('k_attend','k_vocab','k_select_lstm1','k_lstm2','k_select_beam','k_backtrack','k_lstm1')))/calls/1e3
t=sum(float(r['TotalDurationNs']) for r in rows)/calls/1e3
print("   per call: GEMM kernels %.0f us, step pointwise kernels %.0f us, all kernels %.0f us" % (g, p, t))
for r in rows[:8]: print("   %-56s calls %5s avg %7.1f us" % (r['Name'][:56], r['Calls'], float(r['AverageNs'])/1e3))
PY
done 2>&1 | tee $OUT/half_batch_partition_budget.txt
find $OUT -name "*kernel_trace.csv" -delete
