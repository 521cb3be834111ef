set -x
# One round of evidence for profiles/ (run on the GPU box):
#   gpurun -- 'bash tools/profile_round.sh r03h'
# kernel trace of the bench command, the two PMC passes (one counter each, --kernel-trace only) that
# profiles/traffic.json is reduced from, MFMA counters of the f32 head, micro-benchmarks, bench lines.
TAG=${1:-r06z}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-depth72"  # (the step and its probes only: no backbone variants)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- $BENCH > $O/prof_kt.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_kt $O/${TAG}_kernel_trace_bench_f32.md
PMC="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-depth72"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_fetch -o f -- $PMC > $O/prof_fetch.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_fetch $O/${TAG}_pmc_fetch_size.md --ours-only
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_write -o w -- $PMC > $O/prof_write.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_write $O/${TAG}_pmc_write_size.md --ours-only
python $R/tools/pmc_traffic.py /tmp/prof_fetch /tmp/prof_write --tag $TAG --command "bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-depth72" --out $O/${TAG}_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/prof_mfma -o m -- python $R/tools/_pmc_head.py 64 f32 > $O/prof_mfma.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_mfma $O/${TAG}_pmc_head_mfma.md --ours-only
# the sampler's issue / wait / LDS counters (two passes of 8 SQ counters)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_w1 -o a -- python $R/tools/experiments/pmc_warp.py > /dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_w1 $O/${TAG}_pmc_warp_insts.md --ours-only
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/prof_w2 -o b -- python $R/tools/experiments/pmc_warp.py > /dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_w2 $O/${TAG}_pmc_warp_cycles.md --ours-only
# K9 (detector pre-processing, streaming kernel): issue / wait / LDS counters
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_d1 -o a -- python $R/tools/experiments/pmc_detector.py > /dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_d1 $O/${TAG}_pmc_detector_insts.md --ours-only
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/prof_d2 -o b -- python $R/tools/experiments/pmc_detector.py > /dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_d2 $O/${TAG}_pmc_detector_cycles.md --ours-only
cd $R
python tools/microbench.py > $O/${TAG}_microbench.jsonl 2>/dev/null
python tools/experiments/detector_sizes.py > $O/${TAG}_detector_sizes.jsonl 2>/dev/null
timeout 400 python tools/experiments/head_sweep.py quick > $O/${TAG}_head_sweep.jsonl 2>/dev/null
python tools/experiments/fused_vs_unfused.py 2>/dev/null | grep fused > $O/${TAG}_fused_vs_library.txt
python tools/experiments/nhwc_decode_time.py > $O/${TAG}_nhwc_decode.txt 2>/dev/null
cp $O/${TAG}_traffic.json profiles/traffic.json   # (the labelled fallback of bench.py's live PMC passes)
python bench.py > $O/${TAG}_bench_f32.json 2> $O/${TAG}_bench.err
python bench.py --precision f16 --no-cpu-baseline > $O/${TAG}_bench_f16.json 2>> $O/${TAG}_bench.err
python bench.py --config 2 --no-cpu-baseline --steps 10 > $O/${TAG}_bench_config2.json 2>> $O/${TAG}_bench.err
python bench.py --config 3 --no-cpu-baseline > $O/${TAG}_bench_config3.json 2>> $O/${TAG}_bench.err
python bench.py --config 4 --no-cpu-baseline --steps 10 > $O/${TAG}_bench_config4.json 2>> $O/${TAG}_bench.err
MTR_BENCH_SHARED_DEVICE=1 python bench.py --gpus 2 --steps 10 --quick > $O/${TAG}_bench_gpus2_shared_device.json 2>> $O/${TAG}_bench.err
tail -3 $O/prof_kt.log; head -c 2500 $O/${TAG}_bench_f32.json; echo; head -c 1200 $O/${TAG}_bench_f16.json
# one-rank RCCL: the all-gather eager behind the step and captured inside the step's graph
python bench.py --quick --force-collective --step pipeline > $O/${TAG}_bench_rccl_one_rank.json 2>> $O/${TAG}_bench.err
python bench.py --quick --force-collective --graph-gather > $O/${TAG}_bench_rccl_one_rank_graph_gather.json 2>> $O/${TAG}_bench.err
python bench.py --quick --force-collective > $O/${TAG}_bench_rccl_one_rank_api_step.json 2>> $O/${TAG}_bench.err
python tools/head16_ab.py > $O/${TAG}_head16_ab.jsonl 2>/dev/null
# round 5: the backbone's run-to-run stability with and without the deterministic pin; the f32 depth sweep
python tools/experiments/backbone_determinism_probe.py > $O/${TAG}_backbone_determinism.jsonl 2>/dev/null
python tools/experiments/r05_head_probe.py dsweep > $O/${TAG}_f32_depth_sweep_fused_vs_library.jsonl 2>/dev/null
