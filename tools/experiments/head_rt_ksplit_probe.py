"""f32 head at configs[1] (B = 64, C = 1280, 8x8, J = 17: 10 row tiles per crop on 256 CUs = 2.5 tiles per CU):
is a K-HALVED 5 + 5 tile layout -- four workgroups per crop = two row halves x two channel halves, 5 tiles x 20
stages each, partial logits exchanged and merged -- worth building against the shipped 3 + 3 + 3 + 1 (VERDICT r5
next #3)?  Measured WITHOUT writing the kernel: a launch of 128 pseudo-crops of C = 640 with 5-tile blocks
(rt_tiles 5) has exactly the workgroups such a layout would run -- 256 workgroups of 5 tiles x 20 stages, the
crop's half of the channels each -- minus the exchange of the partial logits (80 rows x 64 positions x 8 B = 41 KB
per workgroup pair through L2) and minus nothing else (its epilogue decodes 5 tiles, as the merging side would).
So  t(5+5) >= t(proxy) + t(exchange);  the exchange is priced from MI355X_MICROARCH.md's hand-off table
(handoff-flag + handoff-payload: a fresh 41 KB slot read by the partner costs >= 3 us; a second launch instead
costs a boundary, 1.5 - 1.9 us, plus the merge kernel).  One JSON line per variant -> stdout."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import graph_time  # noqa: E402
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402


def timed(B, C, H, W, J, D, rounds=3, **opts):
    g = torch.Generator(device='cuda').manual_seed(5)
    cfg = MetrabsConfig(depth=D, proc_side=H * 32)
    feat = torch.randn(B, C, H, W, device='cuda', generator=g)
    w = torch.randn(J * (1 + D), C, device='cuda', generator=g) * 0.02
    b = torch.randn(J * (1 + D), device='cuda', generator=g) * 0.1
    packed = kernels.head_pack_weights(w, b, J, D)
    kernels.head_fused(feat, packed, C, J, cfg, **opts)
    us = [graph_time([lambda: kernels.head_fused(feat, packed, C, J, cfg, **opts)] * 20, 5) * 1e6 for _ in range(rounds)]
    plan = kernels.head_plan(B, C, H, W, J, D, **opts)
    return dict(us=round(min(us), 2), us_runs=[round(u, 2) for u in us], kernel=plan['kernel'],
                tiles_per_workgroup=plan['tiles_per_workgroup'], workgroups=plan['workgroups'])


def main():
    J, D = 17, 8
    flops = 2.0 * 1280 * J * (1 + D) * 64 * 64
    out = []
    for name, args, opts in (
            ('shipped: B 64, C 1280, library plan (3+3+3+1 tiles, 40 stages)', (64, 1280, 8, 8), {}),
            ('proxy of the K-halved 5+5 layout: B 128 pseudo-crops of C 640, 5-tile blocks (256 workgroups x 5 tiles x 20 stages), no exchange', (128, 640, 8, 8), dict(rt_tiles=5)),
            ('the same with the loader wave forced', (128, 640, 8, 8), dict(rt_tiles=5, rt_loader=2)),
            ('reference points: B 64, C 1280, 5-tile blocks (128 workgroups x 5 tiles x 40 stages)', (64, 1280, 8, 8), dict(rt_tiles=5)),
            ('B 64, C 640, library plan (what halving K alone buys a 3-tile block)', (64, 640, 8, 8), {}),
            # configs[2]: 12x12 maps, 32 crops per GPU
            ('configs[2] shipped: B 32, C 1280, 12x12, library plan', (32, 1280, 12, 12), {}),
            ('configs[2] proxy K-halved: B 64 pseudo-crops of C 640, 12x12, library plan', (64, 640, 12, 12), {}),
            # the metric string's 72 depth bins
    ):
        B, C, H, W = args
        r = timed(B, C, H, W, J, D, **opts)
        r.update(case=name, B=B, C=C, map=[H, W], options=opts)
        if C == 1280 and (H, W) == (8, 8):
            r['frac_of_157p3TF'] = round(flops / (r['us'] * 1e-6) / 157.3e12, 4)
        print(json.dumps(r), flush=True)
    for name, args, opts in (
            ('D 72 shipped fused: B 64, C 1280, 8x8, 72 bins', (64, 1280, 8, 8), {}),
            ('D 72 proxy K-halved: B 128 pseudo-crops of C 640', (128, 640, 8, 8), {}),
    ):
        B, C, H, W = args
        r = timed(B, C, H, W, J, 72, **opts)
        r.update(case=name, B=B, C=C, map=[H, W], options=opts, D=72)
        print(json.dumps(r), flush=True)


if __name__ == '__main__':
    with torch.inference_mode():
        main()
