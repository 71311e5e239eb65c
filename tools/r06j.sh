#!/bin/bash
# round 6: (1) whole-tile flush for the last piece of a stream-K range (gemm_h2a.h), (2) same-address atomics skipped when the slot already holds more
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06j; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_gemm_fuzz.py tests/test_gpu_headline.py tests/test_gpu_parity.py tests/test_gpu_train.py tests/test_gpu_traj.py tests/test_gpu_flip_rate.py tests/test_gpu_h2.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -6 | tee $OUT/gate.txt
for i in 1 2; do GEMM_REPACK=1 GEMM_NOCHECK=1 timeout 200 tools/gemm_bench 500 256 4 5400 1 2>&1 | grep "A5\|A6\|D5\|C6\|now A5"; done | tee $OUT/repack.txt
for w in beam5 xe; do for r in 1 2; do timeout 400 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu --no-secondary --no-alt 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w rep $r %.0f %s %.3f ms' % (d['value'], d['unit'], d['ms_per_step']))"; done; done | tee $OUT/bench.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats_xe -- python3 $GRAFT_REPO_ROOT/bench.py --workload xe --steps 5 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/$OUT/stats_xe.log 2>&1)
f=$(find $OUT/stats_xe -name "*kernel_stats.csv"); grep "k_bwd_head\|k_bwd_mid\|k_bwd_tail\|k_attend_bwd\|k_row_l1_max\|k_absmax\|k_dlogits" $f | cut -c1-60,150-230 | tee $OUT/xe_kernels.txt
find $OUT -name "*kernel_trace.csv" -delete
