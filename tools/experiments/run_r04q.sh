mkdir -p gpurun_out
( time python __graft_entry__.py smoke ) > gpurun_out/r04q_smoke.log 2>&1
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > gpurun_out/r04q_tests.log 2>&1
( time python bench.py > gpurun_out/r04q_bench_default.json 2> gpurun_out/r04q_bench_default.err ) 2> gpurun_out/r04q_bench_time.log
tail -3 gpurun_out/r04q_smoke.log; tail -3 gpurun_out/r04q_tests.log; cat gpurun_out/r04q_bench_time.log; head -c 200 gpurun_out/r04q_bench_default.json
