#!/bin/bash
# One gpurun call: GPU tests, the default bench line, kernel stats and PMC passes.  usage: tools/gpu_round.sh <tag> [what...]
TAG=${1:-r02_a}; shift
WHAT=${@:-tests bench stats pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for w in $WHAT; do
case $w in
smoke)
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log ;;
tests)
  timeout 1500 python -m pytest tests -m gpu -q -rA --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -40 $OUT/pytest.log ;;
bench)
  timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; cat $OUT/bench_default.json ;;
greedy)
  timeout 300 python bench.py --workload greedy --no-cpu > $OUT/bench_greedy.json 2> $OUT/bench_greedy.err; cat $OUT/bench_greedy.json ;;
xe)
  timeout 300 python bench.py --workload xe --no-cpu > $OUT/bench_xe.json 2> $OUT/bench_xe.err; cat $OUT/bench_xe.json ;;
scst)
  timeout 300 python bench.py --workload scst --no-cpu > $OUT/bench_scst.json 2> $OUT/bench_scst.err; cat $OUT/bench_scst.json ;;
bf16)
  for w in beam5 greedy xe; do timeout 300 python bench.py --workload $w --dtype bf16 --no-cpu --no-secondary > $OUT/bench_${w}_bf16.json 2> $OUT/bench_${w}_bf16.err; cat $OUT/bench_${w}_bf16.json; done
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats_beam5_bf16 -- python3 $GRAFT_REPO_ROOT/bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu --no-secondary --no-alt > $GRAFT_REPO_ROOT/$OUT/stats_beam5_bf16.log 2>&1)
  for f in $(find $OUT/stats_beam5_bf16 -name "*kernel_stats.csv"); do head -12 $f; done
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc16_$c -- python3 $GRAFT_REPO_ROOT/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu --no-secondary --no-alt > $GRAFT_REPO_ROOT/$OUT/pmc16_$c.log 2>&1)
  done
  python tools/hbm_traffic.py $OUT/pmc16_FETCH_SIZE $OUT/pmc16_WRITE_SIZE $OUT/gemm_bf16_hbm_traffic.json gemm_nt_bf16w
  find $OUT -name "*kernel_trace.csv" -size +20M -delete ;;
f32)
  for w in beam5 greedy xe; do timeout 300 python bench.py --workload $w --dtype f32 --no-cpu --no-secondary --no-alt > $OUT/bench_${w}_f32.json 2> $OUT/bench_${w}_f32.err; cat $OUT/bench_${w}_f32.json; done ;;
idx)
  for w in beam5idx xeidx; do timeout 300 python bench.py --workload $w --no-cpu --no-secondary --no-alt > $OUT/bench_${w}.json 2> $OUT/bench_${w}.err; cat $OUT/bench_${w}.json; done ;;
b13)
  timeout 300 python bench.py --batch 13 --no-cpu --no-secondary --no-alt > $OUT/bench_beam5_batch13.json 2> $OUT/bench_beam5_batch13.err; cat $OUT/bench_beam5_batch13.json
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats_batch13 -- python3 $GRAFT_REPO_ROOT/bench.py --batch 13 --steps 20 --warmup 3 --no-cpu --no-secondary --no-alt > $GRAFT_REPO_ROOT/$OUT/stats_batch13.log 2>&1)
  find $OUT -name "*kernel_trace.csv" -size +20M -delete ;;
stats)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats_beam5 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-alt > $GRAFT_REPO_ROOT/$OUT/stats_beam5.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats_xe -- python3 $GRAFT_REPO_ROOT/bench.py --workload xe --steps 5 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/$OUT/stats_xe.log 2>&1)
  find $OUT -name "*kernel_stats.csv" | head; for f in $(find $OUT -name "*kernel_stats.csv"); do echo $f; head -25 $f; done
  # keep only the stats (traces are large)
  find $OUT -name "*kernel_trace.csv" -size +20M -delete ;;
pmc)
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-secondary --no-alt > $GRAFT_REPO_ROOT/$OUT/pmc_$c.log 2>&1)
  done
  python tools/hbm_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/gemm_f16x2_hbm_traffic.json gemm_nt_h2a_kernel
  # XE step: the same two passes over the training workload (the f16x2 kernels of its forward and backward GEMMs together)
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmcxe_$c -- python3 $GRAFT_REPO_ROOT/bench.py --workload xe --steps 2 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/$OUT/pmcxe_$c.log 2>&1)
  done
  python tools/hbm_traffic.py $OUT/pmcxe_FETCH_SIZE $OUT/pmcxe_WRITE_SIZE $OUT/f16x2_xe_step_hbm_traffic.json gemm_nt_h2 "--workload xe --steps 2 --warmup 1 --no-cpu"
  python tools/hbm_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/attend_hbm_traffic.json k_attend
  python tools/hbm_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/vocab_hbm_traffic.json k_vocab
  find $OUT -name "*kernel_trace.csv" -size +20M -delete ;;
esac
done
du -sh $OUT
