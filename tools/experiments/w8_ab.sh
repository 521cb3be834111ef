python -m pytest tests/test_gpu_head.py tests/test_gpu_fuzz.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -3
for v in 1 0; do echo "W8=$v"; MTR_HEAD_W8=$v python tools/microbench.py head 2>&1 | grep kernel | head -7 | cut -c1-130; done
