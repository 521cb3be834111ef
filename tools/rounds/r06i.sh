R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_head.py tests/test_gpu_parity_gates.py tests/test_gpu_predict_multi.py -q -m gpu -x 2>&1 | tail -2
timeout 600 python tools/experiments/head16_pp_probe.py quick > $O/r06i_head16_nchw_rot.jsonl 2>/dev/null
cat $O/r06i_head16_nchw_rot.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'][0], d['shape'][2], 'nhwc' if d['nhwc'] else 'nchw', d['opts'], d['us'], d['bit_equal_to_early_copies'])"
cd /tmp; export TMPDIR=/tmp
PMC_DMA=3 timeout 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/lds_r -o p -- python $R/tools/_pmc_head.py 256 f16 122 12 > /dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/lds_r $O/r06i_lds_dma3_nchw_after.md --ours-only > /dev/null 2>&1
grep -E "head_fused16" $O/r06i_lds_dma3_nchw_after.md | cut -c1-50,150-230
