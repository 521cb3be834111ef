"""A/B of the 16-bit fused head's staging variants on MI355X: dma_staging 1 (four waves copy and multiply)
vs 2 (four MFMA waves + a loader wave), per joint groups per workgroup, against the library pair; every
variant checked bit-equal to the default.  One JSON line per (shape, variant) -> stdout."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import graph_time  # noqa: E402
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402

SHAPES = [  # B, C, J, D, H, W, dtype
    (32, 1280, 122, 8, 12, 12, torch.float16), (256, 1280, 122, 8, 12, 12, torch.float16),
    (64, 1280, 17, 8, 8, 8, torch.float16), (1024, 1280, 17, 8, 8, 8, torch.float16),
    (64, 1280, 17, 8, 8, 8, torch.bfloat16), (256, 2048, 24, 8, 8, 8, torch.float16),
    (32, 1280, 17, 8, 12, 12, torch.float16), (320, 1280, 17, 8, 8, 8, torch.float16),
    (64, 1280, 17, 8, 16, 16, torch.float16)]


def main():
    g = torch.Generator(device='cuda').manual_seed(3)
    for B, C, J, D, H, W, dt in SHAPES:
        for nhwc in (False, True):
            cfg = MetrabsConfig(depth=D, proc_side=max(H, W) * 32)
            feat = torch.randn(B, C, H, W, device='cuda', generator=g).to(dt)
            if nhwc:
                feat = feat.contiguous(memory_format=torch.channels_last)
            w = torch.randn(J * (1 + D), C, device='cuda', generator=g) * 0.02
            b = torch.randn(J * (1 + D), device='cuda', generator=g) * 0.1
            packed = kernels.head_pack_weights(w, b, J, D, dt)
            base = kernels.head_fused(feat, packed, C, J, cfg)
            plan = kernels.head_plan(B, C, H, W, J, D, dt, nhwc)
            flops = 2.0 * C * J * (1 + D) * H * W * B
            for opts in ([dict()] + [dict(dma_staging=s, groups_per_workgroup=gp) for s in (1, 3) for gp in (0, 1, 2, 3)]):
                try:
                    out = kernels.head_fused(feat, packed, C, J, cfg, **opts)
                except RuntimeError as e:
                    print(json.dumps(dict(shape=[B, C, J, D, H, W], dtype=str(dt), nhwc=nhwc, opts=opts, error=str(e)[:80])))
                    continue
                eq = bool(torch.equal(out[0], base[0]) and torch.equal(out[1], base[1]))
                us = graph_time([lambda: kernels.head_fused(feat, packed, C, J, cfg, **opts)] * 20, 5) * 1e6
                print(json.dumps(dict(shape=[B, C, J, D, H, W], dtype=str(dt).split('.')[-1], nhwc=nhwc, opts=opts,
                                      us=round(us, 2), TF=round(flops / us / 1e6, 1), bit_equal=eq,
                                      default_kernel=plan and plan['kernel'])), flush=True)
            if not nhwc:
                conv = torch.nn.Conv2d(C, J * (1 + D), 1).cuda().to(dt)
                with torch.inference_mode():
                    lib = lambda: kernels.softargmax_decode(conv(feat), J, cfg)
                    lib()
                    us = graph_time([lib] * 20, 5) * 1e6
                print(json.dumps(dict(shape=[B, C, J, D, H, W], dtype=str(dt).split('.')[-1], nhwc=nhwc,
                                      opts='library conv + decode', us=round(us, 2), TF=round(flops / us / 1e6, 1))),
                      flush=True)


if __name__ == '__main__':
    with torch.inference_mode():
        main()
