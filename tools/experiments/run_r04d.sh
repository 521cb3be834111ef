mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_api_graphs.py tests/test_gpu_bench_ranks.py -m gpu -q -s 2>&1 | grep -E "parity|passed|failed|Error|assert" | tail -40 ) > gpurun_out/r04d_tests.log 2>&1
timeout 600 python bench.py --no-pmc --cpu-seconds 5 --no-depth72 > gpurun_out/r04d_bench.json 2> gpurun_out/r04d_bench.err
cat gpurun_out/r04d_tests.log; tail -3 gpurun_out/r04d_bench.err; head -c 400 gpurun_out/r04d_bench.json
