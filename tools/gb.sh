for sc in weak strong; do
 echo "=== gloo self-test: 2 ranks on one GPU, beam5 $sc (+ secondary XE)"
 timeout 600 python bench.py --gpus 2 --backend gloo --scaling $sc --steps 4 --warmup 1 --no-cpu 2>&1 | tail -2 | cut -c1-1500
done
echo "=== gloo self-test: 3 ranks, scst"
timeout 600 python bench.py --gpus 3 --backend gloo --workload scst --scaling strong --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | cut -c1-700
