R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_head.py -q -m gpu -x -k "equals_unfused or two_halves or options_are_validated" 2>&1 | tail -2
python - <<'PY'
import torch, sys
sys.path.insert(0,'.')
from metrabs_amd import kernels
from metrabs_amd.config import MetrabsConfig
g=torch.Generator(device='cuda').manual_seed(3)
C,J,D,H,W=1280,122,8,12,12
cfg=MetrabsConfig(depth=D, proc_side=384)
w=torch.randn(J*9,C,device='cuda',generator=g)*0.02; b=torch.randn(J*9,device='cuda',generator=g)*0.1
for dt in (torch.float16, torch.bfloat16):
  packed=kernels.head_pack_weights(w,b,J,D,dt)
  for B in (5, 32, 67):
    feat=torch.randn(B,C,H,W,device='cuda',generator=g).to(dt)
    for f in (feat, feat.contiguous(memory_format=torch.channels_last)):
        base=kernels.head_fused(f,packed,C,J,cfg,dma_staging=3)
        for gp in (1,2):
            o=kernels.head_fused(f,packed,C,J,cfg,dma_staging=7,groups_per_workgroup=gp)
            assert torch.equal(o[0],base[0]) and torch.equal(o[1],base[1]), (dt,B,gp)
print('tight variants bit-equal')
PY
timeout 900 python tools/experiments/head16_pp_probe.py quick > $O/r06k_head16_tight.jsonl 2>/dev/null
cat $O/r06k_head16_tight.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'][0], d['shape'][2], 'nhwc' if d['nhwc'] else 'nchw', d['opts'], d['us'], d['bit_equal_to_early_copies'])"
