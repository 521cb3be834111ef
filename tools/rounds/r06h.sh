R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python tools/experiments/nhwc_decode_ab.py run > $O/r06h_nhwc_decode_ab.jsonl 2> $O/r06h_nhwc.err; echo "nhwc ab rc=$?"
python - <<'PY'
import json, collections
best=collections.defaultdict(lambda: 1e9); sha=collections.defaultdict(set)
for l in open('gpurun_out/r06h_nhwc_decode_ab.jsonl'):
    d=json.loads(l); k=(tuple(d['shape']),d['dtype'],d['variant']); best[k]=min(best[k],d['us']); sha[(tuple(d['shape']),d['dtype'])].add(d['sha256_16'])
for k in sorted(best): print(k, best[k])
print('bit-identical across variants:', all(len(v)==1 for v in sha.values()))
PY
( time python bench.py --no-pmc --no-depth72 --no-api-path --no-decode-roofline > $O/r06h_bench_cpu_baseline.json 2> $O/r06h_bench.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('$O/r06h_bench_cpu_baseline.json')); print(json.dumps(d['cpu_baseline'])[:1500]); print({k:v for k,v in d.items() if k.endswith('_error')})"
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py -q -m gpu -k failing_probe --durations=3 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_decode_recon.py tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -2
