cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d /tmp/ph1 -o a -- python $R/tools/_pmc_head.py 64 f32 > /tmp/ph1.log 2>&1 || tail -3 /tmp/ph1.log
python $R/tools/rocprof_summary.py /tmp/ph1 $R/gpurun_out/r06zc_pmc_head_insts.md --ours-only
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d /tmp/ph2 -o a -- python $R/tools/_pmc_head.py 64 f32 > /tmp/ph2.log 2>&1 || tail -3 /tmp/ph2.log
python $R/tools/rocprof_summary.py /tmp/ph2 $R/gpurun_out/r06zc_pmc_head_cycles.md --ours-only
