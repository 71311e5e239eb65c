python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for r in 0 256 512; do
 for w in greedy xe beam5; do
  echo "=== R16_MAX=$r $w"; VSR_GEMM_R16_MAX=$r timeout 300 python bench.py --workload $w --no-cpu --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'], d['roofline']['gemm_share_of_wall'])"
 done
done
