"""configs[4] 16-bit head (J = 122, 12x12, f16): the two-halves kernel (dma_staging 6, csrc/head_pp.hip) against the
shipped default, the weights-in-registers kernel (4) and -- bit for bit -- each other, at 32 / 64 / 256 / 1024 crops,
both layouts; plus J = 17 and J = 40 on 12x12.  Variants alternated on one box, launches inside a replayed HIP graph.
One JSON line per (shape, layout, variant) -> stdout."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import graph_time  # noqa: E402
if os.environ.get('MTR_PROBE_LIB'):   # a developer build of the library (tools/experiments/variant_lib.py) instead of the product's
    from metrabs_amd import _lib  # noqa: E402
    _lib.load(os.environ['MTR_PROBE_LIB'])
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402

SHAPES = [(32, 1280, 122, 8, 12, 12), (64, 1280, 122, 8, 12, 12), (256, 1280, 122, 8, 12, 12),
          (1024, 1280, 122, 8, 12, 12), (256, 1280, 17, 8, 12, 12), (256, 1280, 40, 8, 12, 12)]
if len(sys.argv) > 1 and sys.argv[1] == 'quick':
    SHAPES = SHAPES[:3]


def main():
    g = torch.Generator(device='cuda').manual_seed(3)
    dt = torch.float16
    for B, C, J, D, H, W in SHAPES:
        for nhwc in (False, True):
            cfg = MetrabsConfig(depth=D, proc_side=max(H, W) * 32)
            feat = torch.randn(B, C, H, W, device='cuda', generator=g).to(dt)
            if nhwc:
                feat = feat.contiguous(memory_format=torch.channels_last)
            w = torch.randn(J * (1 + D), C, device='cuda', generator=g) * 0.02
            b = torch.randn(J * (1 + D), device='cuda', generator=g) * 0.1
            packed = kernels.head_pack_weights(w, b, J, D, dt)
            base = kernels.head_fused(feat, packed, C, J, cfg, dma_staging=3)
            flops = 2.0 * C * J * (1 + D) * H * W * B
            variants = [dict(), dict(dma_staging=3), dict(dma_staging=6), dict(dma_staging=4, groups_per_workgroup=4),
                        dict(dma_staging=7, groups_per_workgroup=1), dict(dma_staging=7, groups_per_workgroup=2)]
            times = {i: [] for i in range(len(variants))}
            eq = {}
            for i, opts in enumerate(variants):
                out = kernels.head_fused(feat, packed, C, J, cfg, **opts)
                eq[i] = bool(torch.equal(out[0], base[0]) and torch.equal(out[1], base[1]))
            for rnd in range(3):   # alternate: clock and neighbour effects hit every variant alike
                for i, opts in enumerate(variants):
                    times[i].append(graph_time([lambda: kernels.head_fused(feat, packed, C, J, cfg, **opts)] * 20, 5) * 1e6)
            for i, opts in enumerate(variants):
                us = min(times[i])
                plan = kernels.head_plan(B, C, H, W, J, D, dt, nhwc, **opts)
                print(json.dumps(dict(shape=[B, C, J, D, H, W], nhwc=nhwc, opts=opts, kernel=plan and plan['kernel'],
                                      us=round(us, 2), us_runs=[round(t, 2) for t in times[i]],
                                      frac_of_2p5PF=round(flops / us / 1e6 / 2.5e6, 4), bit_equal_to_early_copies=eq[i])),
                      flush=True)


if __name__ == '__main__':
    with torch.inference_mode():
        main()
