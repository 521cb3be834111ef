mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r04k_tests.log 2>&1
timeout 300 python tools/head16_ab.py > gpurun_out/r04k_head16_ab.jsonl 2>/dev/null
timeout 600 python bench.py --config 4 --no-cpu-baseline --steps 10 --no-pmc > gpurun_out/r04k_bench_config4.json 2> gpurun_out/r04k_bench.err
timeout 600 python bench.py --precision f16 --no-cpu-baseline --no-pmc > gpurun_out/r04k_bench_f16.json 2>> gpurun_out/r04k_bench.err
tail -5 gpurun_out/r04k_tests.log; tail -2 gpurun_out/r04k_bench.err
