"""Stress of the f32 head's split launches with packed last column blocks: many random batches of 12x12 /
20x20 / 8x12 maps, the planned launch against the plain kernel (bit-equal) and repeated for run-to-run
identity, on two streams at once.  Developer tool (run on the GPU box)."""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
g = torch.Generator(device='cuda').manual_seed(11)
bad = 0
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 150
side_streams = [torch.cuda.Stream() for _ in range(2)]
for it in range(n_iter):
    H, W = rng.choice([(12, 12), (12, 12), (20, 20), (8, 12), (28, 28), (16, 10)])
    B = rng.randint(1, 80 if H * W <= 144 else 24)
    J, D = rng.choice([(17, 8), (24, 8), (5, 20), (122, 8)])
    C = 32 * rng.randint(1, 20)
    cfg = MetrabsConfig(depth=D, proc_side=max(H, W) * 32)
    feat = torch.randn(B, C, H, W, device='cuda', generator=g)
    w = torch.randn(J * (1 + D), C, device='cuda', generator=g) * 0.1
    b = torch.randn(J * (1 + D), device='cuda', generator=g)
    packed = kernels.head_pack_weights(w, b, J, D)
    ref = kernels.head_fused(feat, packed, C, J, cfg, rt_k_groups=1, rt_loader=1, rt_split=1, rt_column_blocks=1,
                             workspace=False)
    torch.cuda.synchronize()
    outs = []
    for st in side_streams:  # two launches in flight, each with its own workspace
        with torch.cuda.stream(st):
            outs.append(kernels.head_fused(feat, packed, C, J, cfg))
    torch.cuda.synchronize()
    for o in outs:
        if not (torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1])):
            bad += 1
            print('MISMATCH', (B, C, J, D, H, W), kernels.head_plan(B, C, H, W, J, D), flush=True)
print(f'{n_iter} shapes, {bad} mismatches')
sys.exit(1 if bad else 0)
