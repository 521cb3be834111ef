R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt16 -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision f16 > $R/gpurun_out/prof_kt16.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_kt16 $R/gpurun_out/r01h_kernel_trace_bench_f16.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f16a -o f -- python $R/tools/_pmc_head.py 1024 f16 > $R/gpurun_out/prof_f16a.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_f16a $R/gpurun_out/r01h_pmc_head16_fetch_b1024.md --ours-only
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/prof_f16b -o m -- python $R/tools/_pmc_head.py 256 f16 122 12 > $R/gpurun_out/prof_f16b.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_f16b $R/gpurun_out/r01h_pmc_head16_mfma_j122.md --ours-only
cd $R
python tools/microbench.py > gpurun_out/r01h_microbench.jsonl 2>/dev/null
python tools/experiments/fused_vs_unfused.py > gpurun_out/r01h_fused_vs_library.txt 2>/dev/null
python bench.py --precision f16 --no-cpu-baseline > gpurun_out/r01h_bench_f16.json 2> gpurun_out/r01h_bench.err
python bench.py > gpurun_out/r01h_bench_f32.json 2>> gpurun_out/r01h_bench.err
grep head_fused gpurun_out/r01h_microbench.jsonl | cut -c1-200; cat gpurun_out/r01h_fused_vs_library.txt; head -c 900 gpurun_out/r01h_bench_f16.json; echo; head -c 400 gpurun_out/r01h_bench_f32.json
