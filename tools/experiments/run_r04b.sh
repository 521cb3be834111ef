mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 ) > gpurun_out/r04b_tests.log 2>&1
timeout 900 python bench.py --cpu-seconds 10 > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err
tail -5 gpurun_out/r04b_tests.log; tail -3 gpurun_out/r04b_bench.err; head -c 600 gpurun_out/r04b_bench.json
