"""Library pair (1x1 conv as the library's GEMM / conv + mtr_softargmax_decode) against the fused head kernel by feature LAYOUT: the
static rule kernels.head_auto_choice was read off NCHW sweeps; channels_last features take another library kernel.  One JSON
line per (dtype, layout, D, map side, B).  Run on the GPU box."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from metrabs_amd import kernels
from metrabs_amd.config import MetrabsConfig
from tools.microbench import timeit

J, C = 17, 1280
CASES = [(torch.float32, D, 8) for D in (8, 16, 24, 32, 48, 72, 80)] + [(torch.float32, 8, s) for s in (12, 16, 20, 24, 32)] + \
        [(dt, 8, s) for dt in (torch.float16, torch.bfloat16) for s in (16, 20, 24)] + [(torch.float16, 72, 8)]
for dt, D, side in CASES:
    for B in (64, 256):
        cfg = MetrabsConfig(depth=D, proc_side=side * 32)
        g = torch.Generator(device='cuda').manual_seed(0)
        w = torch.randn(J * (1 + D), C, 1, 1, device='cuda', generator=g) * 0.02
        bias = torch.randn(J * (1 + D), device='cuda', generator=g) * 0.1
        if not kernels.head_fused_supported(C, J, D, side, side, False, dt):
            continue
        packed = kernels.head_pack_weights(w.reshape(-1, C), bias, J, D, dt)
        for cl in (False, True):
            feat = torch.randn(B, C, side, side, device='cuda', generator=g).to(dt)
            wt = w.to(dt)
            if cl:
                feat = feat.contiguous(memory_format=torch.channels_last)
                wt = wt.contiguous(memory_format=torch.channels_last)
            bt = bias.to(dt)
            with torch.inference_mode():
                t_pair = min(timeit(lambda: kernels.softargmax_decode(F.conv2d(feat, wt, bt), J, cfg)) for _ in range(2))
                t_lin = float('nan')
                if not cl:   # NCHW: logits[b] = W [N, C] @ X[b] [C, HW] as one strided-batched GEMM
                    w2d = wt.reshape(-1, C).contiguous()
                    def bmm():
                        lg = torch.matmul(w2d, feat.view(B, C, side * side)) + bt[None, :, None]
                        return kernels.softargmax_decode(lg.view(B, -1, side, side), J, cfg)
                    t_lin = min(timeit(bmm) for _ in range(2))
                if cl:   # channels_last features ARE a [B H W, C] matrix: the 1x1 conv as a plain GEMM (F.linear), NHWC logits
                    w2d = wt.reshape(-1, C).contiguous()
                    def lin():
                        lg = F.linear(feat.permute(0, 2, 3, 1).reshape(-1, C), w2d, bt)
                        return kernels.softargmax_decode(lg.view(B, side, side, -1).permute(0, 3, 1, 2), J, cfg)
                    t_lin = min(timeit(lin) for _ in range(2))
                try:
                    t_fused = min(timeit(lambda: kernels.head_fused(feat, packed, C, J, cfg)) for _ in range(2))
                except Exception as e:  # (needs a workspace: 16-bit row-tile kernel on NCHW)
                    t_fused = float('nan')
            print(json.dumps(dict(dtype=str(dt).split('.')[-1], D=D, side=side, B=B, channels_last=cl, pair_us=round(t_pair * 1e6, 1),
                                  fused_us=round(t_fused * 1e6, 1), linear_pair_us=round(t_lin * 1e6, 1), fused_over_pair=round(t_fused / t_pair, 3),
                                  rule_says_fused=bool(kernels.head_auto_choice(C, J, D, side, side, cl, dt)))), flush=True)
            del feat
