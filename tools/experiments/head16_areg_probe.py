"""The weights-in-registers 16-bit head kernel (csrc/head_areg.hip, dma_staging 4, 2 - 4 joint groups = waves per
workgroup) against the early-copies kernel (3) and the library's own choice, bit-equality included.  One JSON
line per (shape, layout, options)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import graph_time  # noqa: E402
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402


def main():
    g = torch.Generator(device='cuda').manual_seed(3)
    shapes = [(32, 1280, 122, 8, 12, 12, torch.float16), (64, 1280, 122, 8, 12, 12, torch.float16),
              (128, 1280, 122, 8, 12, 12, torch.float16), (256, 1280, 122, 8, 12, 12, torch.float16),
              (1024, 1280, 122, 8, 12, 12, torch.float16), (256, 1280, 17, 8, 12, 12, torch.bfloat16),
              (64, 1280, 24, 8, 10, 10, torch.float16), (512, 1280, 40, 8, 10, 10, torch.float16)]
    for B, C, J, D, H, W, dt in shapes:
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
            for opts in (dict(), dict(dma_staging=3, groups_per_workgroup=2), dict(dma_staging=4, groups_per_workgroup=2),
                         dict(dma_staging=4, groups_per_workgroup=3), dict(dma_staging=4, groups_per_workgroup=4)):
                out = kernels.head_fused(feat, packed, C, J, cfg, **opts)
                eq = bool(torch.equal(out[0], base[0]) and torch.equal(out[1], base[1]))
                us = graph_time([lambda: kernels.head_fused(feat, packed, C, J, cfg, **opts)] * 20, 5) * 1e6
                print(json.dumps(dict(shape=[B, C, J, D, H, W], dtype=str(dt).split('.')[-1], nhwc=nhwc, opts=opts,
                                      us=round(us, 2), TF=round(flops / us / 1e6, 1), bit_equal=eq)), flush=True)


if __name__ == '__main__':
    with torch.inference_mode():
        main()
