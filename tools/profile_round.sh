set -x
# One round of evidence for profiles/: kernel trace + PMC passes of the bench step (f32 and f16),
# MFMA counters of the head, micro-benchmarks, bench lines.  Run on the GPU box:
#   gpurun -- 'bash tools/profile_round.sh r01g'
TAG=${1:-r01g}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_kt.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_kt $R/gpurun_out/${TAG}_kernel_trace_bench_f32.md
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt16 -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision f16 > $R/gpurun_out/prof_kt16.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_kt16 $R/gpurun_out/${TAG}_kernel_trace_bench_f16.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $R/gpurun_out/prof_fetch.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_fetch $R/gpurun_out/${TAG}_pmc_fetch_size.md --ours-only
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $R/gpurun_out/prof_write.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_write $R/gpurun_out/${TAG}_pmc_write_size.md --ours-only
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/prof_mfma -o m -- python $R/tools/_pmc_head.py 64 f32 > $R/gpurun_out/prof_mfma.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_mfma $R/gpurun_out/${TAG}_pmc_head_mfma.md --ours-only
# the 16-bit head: HBM bytes and matrix-pipe occupancy at a large launch (B = 1024) and at config 5's shape
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f16a -o f -- python $R/tools/_pmc_head.py 1024 f16 > $R/gpurun_out/prof_f16a.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_f16a $R/gpurun_out/${TAG}_pmc_head16_fetch_b1024.md --ours-only
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/prof_f16b -o m -- python $R/tools/_pmc_head.py 256 f16 122 12 > $R/gpurun_out/prof_f16b.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_f16b $R/gpurun_out/${TAG}_pmc_head16_mfma_j122.md --ours-only
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f16c -o f -- python $R/tools/_pmc_head.py 256 f16 122 12 > $R/gpurun_out/prof_f16c.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_f16c $R/gpurun_out/${TAG}_pmc_head16_fetch_j122.md --ours-only
cd $R
python tools/microbench.py > gpurun_out/${TAG}_microbench.jsonl 2>/dev/null
python tools/experiments/fused_vs_unfused.py > gpurun_out/${TAG}_fused_vs_library.txt 2>/dev/null
python bench.py > gpurun_out/${TAG}_bench_f32.json 2> gpurun_out/${TAG}_bench.err
python bench.py --precision f16 --no-cpu-baseline > gpurun_out/${TAG}_bench_f16.json 2>> gpurun_out/${TAG}_bench.err
tail -3 gpurun_out/prof_kt.log; head -c 1500 gpurun_out/${TAG}_bench_f32.json; echo; head -c 1200 gpurun_out/${TAG}_bench_f16.json
