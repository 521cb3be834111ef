import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from bench import graph_time
from metrabs_amd import kernels
from metrabs_amd.config import MetrabsConfig
g = torch.Generator(device='cuda').manual_seed(3)
C, J, D = 1280, 17, 8
w = torch.randn(J * 9, C, device='cuda', generator=g) * 0.02
b = torch.randn(J * 9, device='cuda', generator=g) * 0.1
packed = kernels.head_pack_weights(w, b, J, D)
for B, side in ((32, 12), (24, 12), (40, 12), (48, 12), (72, 8), (80, 8), (96, 8), (64, 8)):
    cfg = MetrabsConfig(depth=D, proc_side=side * 32)
    feat = torch.randn(B, C, side, side, device='cuda', generator=g)
    row = dict(B=B, side=side, plan=kernels.head_plan(B, C, side, side, J, D))
    base = kernels.head_fused(feat, packed, C, J, cfg)
    for name, opts in (('default', {}), ('old_rule_2tile_paired', dict(rt_loader=1, rt_tiles=2, rt_split=2 if side > 8 else 1))):
        out = kernels.head_fused(feat, packed, C, J, cfg, **opts)
        assert torch.equal(out[0], base[0]) and torch.equal(out[1], base[1])
        row[name] = round(min(graph_time([lambda: kernels.head_fused(feat, packed, C, J, cfg, **opts)] * 20, 5) * 1e6 for _ in range(3)), 2)
    print(json.dumps(row), flush=True)
