# Round 6, first GPU call: the new / changed tests, the sharded-estimator file 20x, the whole -m gpu suite 3x,
# the one-rank RCCL bench line WITH the live PMC passes (a rocprofv3 child while the parent holds a communicator).
#   gpurun --timeout 3000 -- 'bash tools/rounds/r06a_gate.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_sharded_estimator.py tests/test_gpu_predict_multi.py -x -q -s -m gpu > $O/r06a_new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/r06a_summary.txt
python -m pytest tests/test_gpu_head.py -x -q -m gpu -k "default_dispatch" >> $O/r06a_new_tests.log 2>&1; echo "default dispatch rc=$?" | tee -a $O/r06a_summary.txt
python -m pytest tests/test_gpu_bench_ranks.py -x -q -m gpu > $O/r06a_bench_ranks.log 2>&1; echo "bench ranks rc=$?" | tee -a $O/r06a_summary.txt
fails=0
for i in $(seq 1 20); do
  python -m pytest tests/test_gpu_sharded_estimator.py -q -m gpu -s > $O/_sh.log 2>&1 || { fails=$((fails+1)); cp $O/_sh.log $O/r06a_sharded_fail_$i.log; }
  echo "run $i: $(tail -1 $O/_sh.log)" >> $O/r06_gputests_x20.log
  grep "^\[sharded\]" $O/_sh.log | sort | uniq -c | sort -rn | head -3 >> $O/r06_gputests_x20.log
done
echo "sharded x20 failures=$fails" | tee -a $O/r06a_summary.txt $O/r06_gputests_x20.log
for i in 1 2 3; do
  python -m pytest tests -q -m gpu > $O/_full.log 2>&1; echo "full suite run $i rc=$?: $(tail -1 $O/_full.log)" | tee -a $O/r06a_summary.txt $O/r06_gputests_x20.log
  grep -E "FAILED|ERROR" $O/_full.log | head -20 >> $O/r06a_summary.txt
done
python bench.py --force-collective --steps 10 --warmup 2 --no-depth72 --no-api-path --cpu-seconds 5 > $O/r06a_bench_rccl_one_rank_with_pmc.json 2> $O/r06a_bench_rccl.err; echo "rccl+pmc bench rc=$?" | tee -a $O/r06a_summary.txt
python -c "
import json; d=json.load(open('$O/r06a_bench_rccl_one_rank_with_pmc.json')); r=d['roofline']
print('roofline', r['kernel'], r['frac'], 'traffic', r['traffic'], r['traffic_source'][:60]); print({k:v for k,v in d.items() if k.endswith('_error')}); print(d['multi_gpu']['backend'])" | tee -a $O/r06a_summary.txt
cat $O/r06a_summary.txt
