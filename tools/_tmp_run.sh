timeout 600 python -m pytest tests/test_gpu_dp.py -x -q -m gpu 2>&1 | tail -15
