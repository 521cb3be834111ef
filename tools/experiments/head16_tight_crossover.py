import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from bench import graph_time
from metrabs_amd import kernels
from metrabs_amd.config import MetrabsConfig
g = torch.Generator(device='cuda').manual_seed(3)
C, D, H, W = 1280, 8, 12, 12
for J in (122, 60, 40):
    cfg = MetrabsConfig(depth=D, proc_side=384)
    w = torch.randn(J * 9, C, device='cuda', generator=g) * 0.02
    b = torch.randn(J * 9, device='cuda', generator=g) * 0.1
    packed = kernels.head_pack_weights(w, b, J, D, torch.float16)
    for B in ((8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 256) if J == 122 else (32, 64, 128, 256)):
        for nhwc in (False, True):
            feat = torch.randn(B, C, H, W, device='cuda', generator=g).half()
            if nhwc:
                feat = feat.contiguous(memory_format=torch.channels_last)
            row = dict(J=J, B=B, nhwc=nhwc)
            for name, opts in (('default', {}), ('tight_gpw1', dict(dma_staging=7, groups_per_workgroup=1))):
                ts = [graph_time([lambda: kernels.head_fused(feat, packed, C, J, cfg, **opts)] * 20, 5) * 1e6 for _ in range(2)]
                row[name] = round(min(ts), 2)
            row['ratio'] = round(row['tight_gpw1'] / row['default'], 3)
            print(json.dumps(row), flush=True)
