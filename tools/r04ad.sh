#!/bin/bash
# where do the weights of the 100-row launches come from?  fabric-side counters of the greedy workload
OUT=gpurun_out/r04ad; mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmcg_$c -- python3 $GRAFT_REPO_ROOT/bench.py --workload greedy --steps 2 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/$OUT/pmcg_$c.log 2>&1)
done
python tools/hbm_traffic.py $OUT/pmcg_FETCH_SIZE $OUT/pmcg_WRITE_SIZE $OUT/f16x2_greedy_hbm_traffic.json gemm_nt_h2a_kernel "--workload greedy --steps 2 --warmup 1 --no-cpu"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmcg_tcc -- python3 $GRAFT_REPO_ROOT/bench.py --workload greedy --steps 2 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/$OUT/pmcg_tcc.log 2>&1)
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for p in glob.glob("$OUT/pmcg_tcc/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(p)):
        if "gemm_nt_h2a" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c,v in sorted(acc.items()): print("%-24s n=%4d mean %.4g"%(c,len(v),sum(v)/len(v)))
PY
find $OUT -name "*kernel_trace.csv" -size +20M -delete
