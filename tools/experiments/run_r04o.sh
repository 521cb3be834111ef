mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_api_graphs.py tests/test_gpu_rccl.py tests/test_gpu_bench_ranks.py -m gpu -q 2>&1 | tail -4 ) > gpurun_out/r04o_tests.log 2>&1
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py --quick --steps 10 --force-collective --graph-gather > gpurun_out/r04o_gg_$i.json 2> gpurun_out/r04o_gg_$i.err; echo "graph-gather run $i rc=$?" >> gpurun_out/r04o_rc.log
  timeout 300 python bench.py --quick --steps 10 --force-collective > gpurun_out/r04o_api_$i.json 2> gpurun_out/r04o_api_$i.err; echo "api run $i rc=$?" >> gpurun_out/r04o_rc.log
done
timeout 900 python bench.py --cpu-seconds 5 --no-depth72 > gpurun_out/r04o_bench.json 2> gpurun_out/r04o_bench.err
cat gpurun_out/r04o_rc.log; tail -3 gpurun_out/r04o_tests.log; tail -2 gpurun_out/r04o_bench.err
