#!/bin/bash
# PMC passes over the 16-bit fused head at config 5's shape (J=122, 12x12) -- developer probe
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
ARGS="${PMC_ARGS:-256 f16 122 12}"
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc16_$i -o p -- python $R/tools/_pmc_head.py $ARGS > $R/gpurun_out/pmc16_$i.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmc16_$i $R/gpurun_out/pmc16_$i.md --ours-only 2>&1 | tail -2
done
cat $R/gpurun_out/pmc16_*.md | grep -v "^$" | head -80
