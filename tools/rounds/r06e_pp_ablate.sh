# the two-halves kernel's stage budget with parts switched off (MTR_PP_ABLATE: 4 no MFMA, 8 no copies in the loop, 16 no fragment reads)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for k in 8 16 4 24 12 20; do
  echo "# ablate $k" >> $O/r06e_pp_trace_ablations.jsonl
  MTR_PROBE_LIB=$R/tools/experiments/_build/libmtr_pptrace_a$k.so python tools/experiments/head16_pp_trace.py 2>/dev/null | grep -E '"block": 0,' >> $O/r06e_pp_trace_ablations.jsonl
done
python - <<'PY'
import json
k=None
for l in open('gpurun_out/r06e_pp_trace_ablations.jsonl'):
    if l.startswith('#'): k=l.strip(); continue
    d=json.loads(l)
    x,y=d['X']['per_stage_cycles'],d['Y']['per_stage_cycles']
    print(k, 'B',d['B'],'nhwc' if d['nhwc'] else 'nchw','stage',d['X']['stage_total'],'| X compute',x['phase0_work'],'X issue',x['phase1_work'],'| Y issue',y['phase0_work'],'Y compute',y['phase1_work'],'| waits',x['phase0_wait_copies'],y['phase0_wait_copies'],x['phase1_wait_copies'],y['phase1_wait_copies'])
PY
