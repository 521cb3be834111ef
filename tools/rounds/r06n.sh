R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_decode_recon.py tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -4
timeout 900 python tools/experiments/nhwc_decode_ab.py run > $O/r06n_nhwc_quad_ab.jsonl 2> $O/r06n.err
python - <<'PY'
import json, collections
best=collections.defaultdict(lambda: 1e9)
for l in open('gpurun_out/r06n_nhwc_quad_ab.jsonl'):
    d=json.loads(l); k=(tuple(d['shape']),d['dtype'],d['variant']); best[k]=min(best[k],d['us'])
for k in sorted(best): print(k, best[k])
PY
