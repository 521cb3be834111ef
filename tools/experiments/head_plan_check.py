"""f32 head launch plan vs measurement: for a range of batch sizes (8x8 maps, J = 17, D = 8, C = 1280) the
time of the planned launch ('auto'), of every (kernel, tiles per workgroup) choice and of the library
pair, beside the plan's own estimate.  Developer tool (run on the GPU box).

    python tools/experiments/head_plan_check.py [B ...] > gpurun_out/head_plan_check.jsonl
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402
from tools.experiments.head_sweep import timed  # noqa: E402


def main():
    batches = [int(a) for a in sys.argv[1:]] or [96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 448, 512, 640, 768]
    C, J, D, side = 1280, 17, 8, 8
    cfg = MetrabsConfig(depth=D)
    g = torch.Generator(device='cuda').manual_seed(0)
    w = torch.randn(J * (1 + D), C, device='cuda', generator=g) * 0.03
    bias = torch.randn(J * (1 + D), device='cuda', generator=g) * 0.1
    packed = kernels.head_pack_weights(w, bias, J, D, torch.float32)
    for B in batches:
        feat = torch.randn(B, C, side, side, device='cuda', generator=g)
        out = (torch.empty(B, J, 2, device='cuda'), torch.empty(B, J, 3, device='cuda'))
        row = {'B': B}
        plan = kernels.head_plan(B, C, side, side, J, D, torch.float32, False, True)
        row['plan'] = f"{plan['kernel']} x{plan['tiles_per_workgroup']}"
        row['model_us'] = round(plan['model_us'], 1)
        row['auto_us'] = round(timed(lambda: kernels.head_fused(feat, packed, C, J, cfg, out=out, workspace=False)), 1)
        w4 = w.view(-1, C, 1, 1)
        row['library_us'] = round(timed(lambda: kernels.softargmax_decode(F.conv2d(feat, w4, bias), J, cfg)), 1)
        best = None
        for ld in (1, 2):
            for rt in (1, 2, 3, 4, 5):
                t = timed(lambda: kernels.head_fused(feat, packed, C, J, cfg, out=out, workspace=False, rt_loader=ld,
                                                     rt_tiles=rt))
                p = kernels.head_plan(B, C, side, side, J, D, torch.float32, False, True, rt_loader=ld, rt_tiles=rt)
                row[f"{'ld' if ld == 2 else 'plain'}{rt}"] = [round(t, 1), round(p['model_us'], 1)]
                if best is None or t < best[0]:
                    best = (t, f"{'head_rt_ld_kernel' if ld == 2 else 'head_rt_kernel'} x{rt}")
        row['best_us'], row['best'] = round(best[0], 1), best[1]
        row['auto_over_best'] = round(row['auto_us'] / row['best_us'], 3)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
