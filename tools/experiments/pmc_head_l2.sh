# L2-side counters of the f32 head (developer tool; run on the GPU box):
#   gpurun -- 'bash tools/experiments/pmc_head_l2.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 64 1024; do
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum FETCH_SIZE GRBM_GUI_ACTIVE"; do
    n=$(echo $set | cut -d' ' -f1)
    rm -rf /tmp/pmc_l2
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_l2 -o out -- python $R/tools/_pmc_head.py $B f32 > /dev/null 2>&1
    python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for p in glob.glob('/tmp/pmc_l2/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'head_rt' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']:
            acc[(r['Kernel_Name'].split('(')[0][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print('B=$B', k[0], k[1], round(sum(v) / len(v), 1), 'n=%d' % len(v))
PY
  done
done
