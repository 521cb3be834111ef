R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1200 python -m pytest tests/test_gpu_head.py tests/test_gpu_parity_gates.py tests/test_gpu_predict_multi.py tests/test_gpu_baseline_configs.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python tools/experiments/head16_tight_crossover.py 2>/dev/null | grep '"J": 122' > $O/r06m_head16_default_after.jsonl; cat $O/r06m_head16_default_after.jsonl
python bench.py --config 4 --no-cpu-baseline --steps 10 > $O/r06m_bench_config4.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r06m_bench_config4.json')); r=d['roofline']; print('config4', round(d['value'],1), r['kernel'], round(r['frac'],4), round(r['avg_launch_us'],2), r['traffic'])"
