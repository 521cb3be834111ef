# Round 6, second GPU call: the two-halves 16-bit head (bit identity, then timing against the shipped kernels), the L2-hit
# ingest ceiling, the persistent sampler launch, the f32 K-split proxy, the changed tests.
#   gpurun --timeout 2400 -- 'bash tools/rounds/r06b_probes.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_head.py -x -q -m gpu -k "two_halves or equals_unfused or options_are_validated or default_dispatch" > $O/r06b_pp_tests.log 2>&1; echo "pp tests rc=$?: $(tail -1 $O/r06b_pp_tests.log)" | tee -a $O/r06b_summary.txt
timeout 900 python tools/experiments/head16_pp_probe.py > $O/r06b_head16_pp.jsonl 2> $O/r06b_head16_pp.err; echo "pp probe rc=$?" | tee -a $O/r06b_summary.txt
timeout 300 tools/experiments/_build/ingest_probe > $O/r06b_ingest_probe.txt 2>&1; echo "ingest rc=$?" | tee -a $O/r06b_summary.txt
V=r0,persist2,persist4,persist8,persist4r8,persist8pd2
ABLATE_ONLY=$V ABLATE_ROTATE=3 timeout 600 python tools/experiments/ablate_warp.py run > $O/r06b_warp_persist_64.jsonl 2>/dev/null
ABLATE_ONLY=$V ABLATE_ROTATE=3 ABLATE_CROPS=320 ABLATE_AUG=5 timeout 600 python tools/experiments/ablate_warp.py run > $O/r06b_warp_persist_320tta.jsonl 2>/dev/null
echo "warp ablation done" | tee -a $O/r06b_summary.txt
timeout 600 python tools/experiments/head_rt_ksplit_probe.py > $O/r06_head_rt_ksplit.jsonl 2> $O/r06b_ksplit.err; echo "ksplit rc=$?" | tee -a $O/r06b_summary.txt
timeout 900 python -m pytest tests/test_gpu_latent.py tests/test_gpu_predict_multi.py -q -m gpu > $O/r06b_changed_tests.log 2>&1; echo "changed tests rc=$?: $(tail -1 $O/r06b_changed_tests.log)" | tee -a $O/r06b_summary.txt
cat $O/r06b_head16_pp.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'][0], d['shape'][2], 'nhwc' if d['nhwc'] else 'nchw', d['opts'], d['us'], d['frac_of_2p5PF'], d['bit_equal_to_early_copies'])"
cat $O/r06b_warp_persist_64.jsonl $O/r06b_warp_persist_320tta.jsonl; tail -14 $O/r06b_ingest_probe.txt; cat $O/r06_head_rt_ksplit.jsonl | cut -c1-260
cat $O/r06b_summary.txt
