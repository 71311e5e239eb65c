cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_live_forwards.py tests/test_gpu_weight_generation.py tests/test_gpu_configs.py tests/test_gpu_gemm_fuzz.py tests/test_gpu_headline.py -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest.txt; cat $OUT/pytest.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/calib_fetch -- $GRAFT_REPO_ROOT/tools/fetch_calib > $GRAFT_REPO_ROOT/$OUT/calib_fetch.log 2>&1)
python tools/fetch_calib.py $OUT/calib_fetch > $OUT/fetch_size_calibration.txt; cat $OUT/fetch_size_calibration.txt
find $OUT -name "*kernel_trace.csv" -delete
bash tools/ab.sh r06c_ab -w "beam5 greedy xe" -r 2 -s 20 "VSR_XCD_GROUPS=1" "VSR_XCD_GROUPS=0"
