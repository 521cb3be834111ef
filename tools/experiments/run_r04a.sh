mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_api_graphs.py tests/test_gpu_rccl.py tests/test_gpu_pipeline.py "tests/test_gpu_head.py::test_auto_head_path_is_a_static_rule" tests/test_gpu_baseline_configs.py tests/test_gpu_sharded_estimator.py -m gpu -q -s -x 2>&1 | tail -80 ) > gpurun_out/r04a_tests.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s 2>&1 | grep -E "parity|passed|failed|Error" ) > gpurun_out/r04a_e2e.log 2>&1
timeout 900 python bench.py --no-pmc --cpu-seconds 5 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
timeout 300 python bench.py --quick --force-collective > gpurun_out/r04a_bench_rccl.json 2> gpurun_out/r04a_bench_rccl.err
timeout 300 python bench.py --quick --force-collective --graph-gather > gpurun_out/r04a_bench_rccl_graph.json 2> gpurun_out/r04a_bench_rccl_graph.err
tail -5 gpurun_out/r04a_tests.log; tail -3 gpurun_out/r04a_bench.err; tail -3 gpurun_out/r04a_bench_rccl.err
