mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_head.py -m gpu -q -x -k "16bit or options or half or bfloat" 2>&1 | tail -8 ) > gpurun_out/r04i_tests.log 2>&1
timeout 400 python tools/head16_ab.py > gpurun_out/r04i_head16_ab.jsonl 2>gpurun_out/r04i_ab.err
tail -4 gpurun_out/r04i_tests.log; tail -2 gpurun_out/r04i_ab.err
