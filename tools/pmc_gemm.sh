#!/bin/bash
# SQ counters of one gemm_bench configuration: tools/pmc_gemm.sh <tag> <gemm_bench args...>   (env GEMM_PLAN_ALIGNED passes through)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$n -- $GRAFT_REPO_ROOT/tools/gemm_bench "$@" > $OUT/$n.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$OUT/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"][:60]
        if "gemm" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k)
    for c,vals in sorted(v.items()):
        vals=vals[-20:]
        print("   %-32s n=%3d  mean %.4g"%(c,len(vals),sum(vals)/len(vals)))
PY
