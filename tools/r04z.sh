#!/bin/bash
# att_ga in the vocabulary launch (five problems, all-DMA kernel) so that LSTM2 alone fills 256 CUs: parity gate, then A/B
OUT=gpurun_out/r04z; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_headline.py -m gpu -q 2>&1 | tail -6 > $OUT/tests.txt
cat $OUT/tests.txt
echo "(gate result above; timings run regardless in this experiment)"
for rep in 1 2; do for ga in 1 0; do
  echo "== VSR_GA_IN_S6=$ga rep $rep"
  VSR_GA_IN_S6=$ga timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu --no-secondary --no-alt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam5', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  VSR_GA_IN_S6=$ga timeout 300 python bench.py --workload greedy --steps 30 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('greedy', d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $OUT/ga_in_s6_ab.txt
