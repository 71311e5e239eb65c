python -m pytest tests/test_gpu_ssp.py -m gpu -q 2>&1 | tail -5
