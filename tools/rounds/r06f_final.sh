# Round 6, evidence run: the whole -m gpu suite (with durations), the round's profile set (tools/profile_round.sh r06z:
# kernel trace of the bench command, PMC passes, micro-benchmarks, one bench line per BASELINE config), the 16-bit head's
# counters at configs[4]'s shape (MFMA busy, FETCH_SIZE, LDS) at 32 and 256 crops.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests -q -m gpu --durations=25 > $O/r06f_gputests.log 2>&1; echo "gpu suite rc=$?: $(tail -1 $O/r06f_gputests.log)" | tee -a $O/r06f_summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06f_smoke.log 2>&1; echo "smoke rc=$?: $(tail -1 $O/r06f_smoke.log)" | tee -a $O/r06f_summary.txt
( time python bench.py > $O/r06f_bench_default.json 2> $O/r06f_bench_default.err ) 2> $O/r06f_bench_default.time; echo "default bench rc=$? $(grep real $O/r06f_bench_default.time)" | tee -a $O/r06f_summary.txt
bash tools/profile_round.sh r06z > $O/r06f_profile_round.log 2>&1; echo "profile round rc=$?" | tee -a $O/r06f_summary.txt
for B in 32 256; do
  PMC_ARGS="$B f16 122 12" bash tools/experiments/pmc_head16.sh > /dev/null 2>&1
  cat $O/pmc16_1.md $O/pmc16_2.md $O/pmc16_4.md > $O/r06_pmc_head16_j122_b$B.md
done
echo "pmc head16 done" | tee -a $O/r06f_summary.txt
grep -A30 "slowest 25 durations" $O/r06f_gputests.log | head -32
python - <<'PY'
import json
for f in ('r06f_bench_default','r06z_bench_f32','r06z_bench_f16','r06z_bench_config2','r06z_bench_config3','r06z_bench_config4'):
    try:
        d=json.load(open('gpurun_out/%s.json'%f)); r=d['roofline'] or {}
        print(f, round(d['value'],1), round(d['ms_per_step'],3), r.get('kernel'), round(r.get('frac',0),4), r.get('avg_launch_us'), r.get('traffic'), r.get('tie_within_5pct'), {k:v for k,v in d.items() if k.endswith('_error')}, d.get('cpu_baseline') and (round(d['cpu_baseline']['value'],2), d['cpu_baseline']['cores'], d['cpu_baseline'].get('crops_per_s_by_threads')))
    except Exception as e: print(f, 'ERR', e)
PY
cat $O/r06f_summary.txt
