mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_head.py tests/test_gpu_parity_gates.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r04g_tests.log 2>&1
timeout 300 python tools/experiments/head_fixed_vs_stage.py > gpurun_out/r04g_head_fixed_vs_stage.jsonl 2>/dev/null
grep parity gpurun_out/parity_report.jsonl 2>/dev/null | tail -0
tail -6 gpurun_out/r04g_tests.log; grep float16 gpurun_out/r04g_head_fixed_vs_stage.jsonl
