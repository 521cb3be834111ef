set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_kt.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_kt $R/gpurun_out/r01f_kernel_trace.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $R/gpurun_out/prof_fetch.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_fetch $R/gpurun_out/r01f_pmc_fetch.md --ours-only
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $R/gpurun_out/prof_write.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_write $R/gpurun_out/r01f_pmc_write.md --ours-only
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/prof_mfma -o m -- python $R/tools/_pmc_head.py 64 f32 > $R/gpurun_out/prof_mfma.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_mfma $R/gpurun_out/r01f_pmc_head_mfma.md --ours-only
cd $R
python tools/microbench.py > gpurun_out/r01f_microbench.jsonl 2>/dev/null
python bench.py > gpurun_out/r01f_bench_f32.json 2> gpurun_out/r01f_bench.err
tail -3 gpurun_out/prof_kt.log; cat gpurun_out/r01f_bench_f32.json | head -c 1500
