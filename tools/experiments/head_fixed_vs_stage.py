"""Fixed cost vs per-stage cost of the fused heads: the same launch at C = 64 ... 1280 (one JSON line each)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import graph_time  # noqa: E402
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402

CASES = [(256, 122, 12, torch.float16), (32, 122, 12, torch.float16), (64, 17, 8, torch.float16),
         (1024, 17, 8, torch.float16), (64, 17, 8, torch.float32), (1024, 17, 8, torch.float32)]
with torch.inference_mode():
    g = torch.Generator(device='cuda').manual_seed(3)
    for B, J, side, dt in CASES:
        for C in (64, 128, 320, 640, 1280):
            cfg = MetrabsConfig(depth=8, proc_side=side * 32)
            feat = torch.randn(B, C, side, side, device='cuda', generator=g).to(dt)
            w = torch.randn(J * 9, C, device='cuda', generator=g) * 0.02
            b = torch.zeros(J * 9, device='cuda')
            packed = kernels.head_pack_weights(w, b, J, 8, dt)
            kernels.head_fused(feat, packed, C, J, cfg)
            us = graph_time([lambda: kernels.head_fused(feat, packed, C, J, cfg)] * 20, 5) * 1e6
            plan = kernels.head_plan(B, C, side, side, J, 8, dt)
            print(json.dumps(dict(B=B, J=J, side=side, dtype=str(dt).split('.')[-1], C=C, us=round(us, 2),
                                  kernel=plan and plan['kernel'], wgs=plan and plan['workgroups'])), flush=True)
