# PMC passes over the NHWC decode kernels (each counter set in its own run, kernel trace only beside it):
#   bash tools/experiments/pmc_nhwc.sh <tag>   ->  gpurun_out/<tag>_pmc_nhwc_<n>.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmcn_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcn_$i -o out -- python $R/tools/experiments/pmc_nhwc.py > /tmp/pmcn_$i.log 2>&1 || tail -5 /tmp/pmcn_$i.log
  python $R/tools/rocprof_summary.py /tmp/pmcn_$i $R/gpurun_out/$1_pmc_nhwc_$i.md --ours-only > /dev/null 2>&1 || echo "summary $i failed"
done
