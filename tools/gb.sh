for v in 0 1 2 3; do echo "=== x3 exp $v"; GEMM_NOCHECK=1 timeout 120 tools/gemm_bench_x$v 500 256 4 3300 1 2>&1 | tail -7; done
