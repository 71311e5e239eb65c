#!/bin/bash
# round 6: whole-m-group XCD dealing of the k-aligned GEMM plans (balanced) - end-to-end A/B and FETCH_SIZE per launch of the wide kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06d; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_gemm_fuzz.py tests/test_gpu_headline.py -m gpu -q -x 2>&1 | tail -5
bash tools/ab.sh r06d_ab -w "beam5 b13" -r 3 -s 20 "VSR_XCD_GROUPS=1" "VSR_XCD_GROUPS=0"
for v in 1 0; do
  (cd /tmp && VSR_XCD_GROUPS=$v timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/fetch_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-secondary --no-alt > $GRAFT_REPO_ROOT/$OUT/fetch_$v.log 2>&1)
  python3 - <<PY
import csv, glob
acc = {}
for p in glob.glob("$OUT/fetch_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p, newline="")):
        if r["Counter_Name"] == "FETCH_SIZE" and "gemm_nt_h2a" in r["Kernel_Name"]:
            k = r["Kernel_Name"][:48]
            acc.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("VSR_XCD_GROUPS=$v  %-50s launches %4d  FETCH_SIZE %.1f KiB per launch  -> x 2 = %.1f MB at the fabric" % (k, len(v), sum(v) / len(v), 2 * 1024 * sum(v) / len(v) / 1e6))
PY
done | tee $OUT/fetch_per_launch.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +5M -delete
