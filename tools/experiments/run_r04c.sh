mkdir -p gpurun_out
( timeout 300 python tools/head16_ab.py > gpurun_out/r04c_head16_ab.jsonl 2> gpurun_out/r04c_head16_ab.err )
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r04c_tests.log 2>&1
timeout 600 python bench.py --no-pmc --cpu-seconds 5 --no-depth72 > gpurun_out/r04c_bench.json 2> gpurun_out/r04c_bench.err
( MTR_BENCH_SHARED_DEVICE=1 timeout 600 python bench.py --gpus 2 --quick > gpurun_out/r04c_bench_gpus2_shared.json 2> gpurun_out/r04c_bench_gpus2_shared.err )
tail -5 gpurun_out/r04c_tests.log; tail -3 gpurun_out/r04c_bench.err; tail -2 gpurun_out/r04c_head16_ab.err; head -c 300 gpurun_out/r04c_bench_gpus2_shared.json
