python -m pytest tests/test_gpu_bf16.py tests/test_aa_gpu_dp.py -m gpu -q -s 2>&1 | tail -25
for w in beam5 greedy xe; do for dt in f32 bf16; do
  echo "=== $w $dt"; timeout 300 python bench.py --workload $w --dtype $dt --no-cpu --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'], d['roofline']['gemm_share_of_wall'])"
done; done
