R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/p.py <<'PY'
import torch
feat = torch.randn(64, 1280, 8, 8, device='cuda'); w = torch.randn(153, 1280, 1, 1, device='cuda')*0.03; b=torch.zeros(153,device='cuda')
for _ in range(5): y = torch.nn.functional.conv2d(feat, w, b)
feat = torch.randn(1024, 1280, 8, 8, device='cuda')
for _ in range(5): y = torch.nn.functional.conv2d(feat, w, b)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o c -- python /tmp/p.py > /dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_c $R/gpurun_out/conv_trace.md
head -14 $R/gpurun_out/conv_trace.md | cut -c1-200
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_c/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
seen=set()
for r in rows:
    k=r['Kernel_Name'][:60]
    if k in seen: continue
    seen.add(k)
    print(k, '| grid',r.get('Grid_Size_X'),r.get('Grid_Size_Y'),r.get('Grid_Size_Z'),'| wg',r.get('Workgroup_Size_X'),r.get('Workgroup_Size_Y'),'| lds',r.get('LDS_Block_Size'),'| vgpr',r.get('VGPR_Count'),'accum',r.get('Accum_VGPR_Count'),'sgpr',r.get('SGPR_Count'),'| dur',int(r['End_Timestamp'])-int(r['Start_Timestamp']))
PY
