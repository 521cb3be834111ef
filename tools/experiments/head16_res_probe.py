"""The resident-weights 16-bit head kernel (csrc/head_res.hip, dma_staging 5) against the library's own choice:
bit-equality and time, J = 122 on 12x12 (configs[4]) and neighbours.  One JSON line per case -> stdout."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import graph_time  # noqa: E402
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402

SHAPES = [(32, 122, 8, 12, 12, torch.float16), (256, 122, 8, 12, 12, torch.float16), (1024, 122, 8, 12, 12, torch.float16),
          (7, 122, 8, 12, 12, torch.bfloat16), (64, 122, 8, 10, 10, torch.float16), (64, 24, 8, 12, 12, torch.float16),
          (37, 122, 8, 8, 12, torch.float16), (256, 17, 8, 12, 12, torch.float16)]
if len(sys.argv) > 1 and sys.argv[1] == 'quick':
    SHAPES = SHAPES[:2]


def main():
    g = torch.Generator(device='cuda').manual_seed(3)
    C = 1280
    for B, J, D, H, W, dt in SHAPES:
        for nhwc in (False, True):
            cfg = MetrabsConfig(depth=D, proc_side=max(H, W) * 32)
            feat = torch.randn(B, C, H, W, device='cuda', generator=g).to(dt)
            if nhwc:
                feat = feat.contiguous(memory_format=torch.channels_last)
            w = torch.randn(J * (1 + D), C, device='cuda', generator=g) * 0.02
            b = torch.randn(J * (1 + D), device='cuda', generator=g) * 0.1
            packed = kernels.head_pack_weights(w, b, J, D, dt)
            base = kernels.head_fused(feat, packed, C, J, cfg)
            flops = 2.0 * C * J * (1 + D) * H * W * B
            for opts in (dict(), dict(dma_staging=5)):
                try:
                    out = kernels.head_fused(feat, packed, C, J, cfg, **opts)
                    torch.cuda.synchronize()
                except RuntimeError as e:
                    print(json.dumps(dict(shape=[B, C, J, D, H, W], nhwc=nhwc, opts=opts, error=str(e)[:120])), flush=True)
                    continue
                eq = bool(torch.equal(out[0], base[0]) and torch.equal(out[1], base[1]))
                err = float((out[1] - base[1]).abs().max())
                us = graph_time([lambda: kernels.head_fused(feat, packed, C, J, cfg, **opts)] * 10, 5) * 1e6
                print(json.dumps(dict(shape=[B, C, J, D, H, W], dtype=str(dt).split('.')[-1], nhwc=nhwc, opts=opts,
                                      us=round(us, 2), TF=round(flops / us / 1e6, 1), frac_mfma=round(flops / us / 1e6 / 2500, 3),
                                      bit_equal=eq, max_abs_diff=err)), flush=True)


if __name__ == '__main__':
    with torch.inference_mode():
        main()
