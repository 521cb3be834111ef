# LDS counters of the 16-bit head kernels at configs[4]'s shape: which reads conflict?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
i=0
for spec in "3 nchw" "3 nhwc" "6 nhwc" "6 nchw" "4 nhwc"; do
  set -- $spec; i=$((i+1))
  PMC_DMA=$1 timeout 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/lds_$i -o p -- python $R/tools/_pmc_head.py 256 f16 122 12 $2 > $O/lds_$i.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/lds_$i $O/r06g_lds_dma$1_$2.md --ours-only > /dev/null 2>&1
  echo "== dma_staging $1 $2"; grep -E "head_fused16" $O/r06g_lds_dma$1_$2.md | grep -E "LDS|WAVE_CYC|avg us|\*\*" | cut -c1-60,150-230
done
tail -3 $O/lds_1.log
