( timeout 900 python -m pytest tests/test_gpu_api_graphs.py tests/test_gpu_sharded_estimator.py -m gpu -q -x 2>&1 | tail -60 ) > gpurun_out/r04l_tests.log 2>&1
