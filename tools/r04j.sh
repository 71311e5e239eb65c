#!/bin/bash
# does the L2 share the weight tile between the m-tiles of one n-tile?  TCC misses / fabric reads per launch of the wide f16x2 kernel at 1, 2, 4 m-tiles
export GEMM_PLAN_ALIGNED=4
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04j2; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for M in 128 256 512; do
 for set in "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum" "FETCH_SIZE"; do
  n=M${M}_$(echo $set | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$n -- $GRAFT_REPO_ROOT/tools/gemm_bench $M 256 4 5200 1 > $OUT/$n.log 2>&1
 done
done
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$OUT/**/*counter_collection.csv",recursive=True):
    M=re.search(r"/M(\d+)_",p).group(1)
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"][:44]
        if "gemm_nt_h2" not in k: continue
        acc[(M,k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()):
    print(k)
    for c,vals in sorted(v.items()):
        vals=vals[-20:]
        print("   %-32s n=%3d  mean %.4g"%(c,len(vals),sum(vals)/len(vals)))
PY
for M in 128 256 512; do grep -h -E "^S[1256]|step GEMMs" $OUT/M${M}_FETCH_SIZE.log | head -5; done
