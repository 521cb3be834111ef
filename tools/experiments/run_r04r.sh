mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_head.py tests/test_gpu_parity_gates.py tests/test_gpu_fuzz.py tests/test_gpu_decode_recon.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r04r_tests.log 2>&1
timeout 300 python tools/experiments/head_fixed_vs_stage.py > gpurun_out/r04r_head_fixed_vs_stage.jsonl 2>/dev/null
timeout 200 python tools/experiments/fused_vs_unfused.py 2>/dev/null | grep fused > gpurun_out/r04r_fused_vs_library.txt
tail -6 gpurun_out/r04r_tests.log; grep float32 gpurun_out/r04r_head_fixed_vs_stage.jsonl; cat gpurun_out/r04r_fused_vs_library.txt
