"""Round-5 head probes on MI355X, one JSON line each:
  dsweep: f32 features, depth bins 8 ... 72: fused row-tile kernel vs library 1x1 conv + decode (for the
         static rule of kernels.head_auto_choice)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import graph_time  # noqa: E402
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402


def dsweep():
    g = torch.Generator(device='cuda').manual_seed(4)
    for B in (64, 1024):
        for D in (8, 16, 24, 32, 40, 48, 56, 63, 64, 72, 80):
            C, J, H, W = 1280, 17, 8, 8
            cfg = MetrabsConfig(depth=D)
            feat = torch.randn(B, C, H, W, device='cuda', generator=g)
            conv = torch.nn.Conv2d(C, J * (1 + D), 1).cuda()
            packed = kernels.head_pack_weights(conv.weight.detach().reshape(J * (1 + D), C), conv.bias.detach(), J, D)
            fused = lambda: kernels.head_fused(feat, packed, C, J, cfg)
            lib = lambda: kernels.softargmax_decode(conv(feat), J, cfg)
            tf = graph_time([fused] * 10, 5) * 1e6
            tl = graph_time([lib] * 10, 5) * 1e6
            print(json.dumps(dict(probe='dsweep', B=B, D=D, fused_us=round(tf, 1), library_us=round(tl, 1),
                                  fused_over_library=round(tf / tl, 3))), flush=True)


if __name__ == '__main__':
    with torch.inference_mode():
        for name in (sys.argv[1:] or ['dsweep']):
            {'dsweep': dsweep}[name]()
