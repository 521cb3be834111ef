"""f32 fused head: every dispatch choice on every shape of DESIGN.md section 3's table, graph-replayed,
beside the library pair (F.conv2d 1x1 + mtr_softargmax_decode), with a bit-equality check of every
choice against the one-K-group / no-loader / no-split kernel.  Developer tool (run on the GPU box).

    python tools/experiments/head_sweep.py [quick] > gpurun_out/head_sweep.jsonl
"""
import itertools
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402

SHAPES = [  # label, B, C, J, D, side, nhwc
    ('B64 8x8', 64, 1280, 17, 8, 8, False),
    ('B64 8x8 nhwc', 64, 1280, 17, 8, 8, True),
    ('B256 8x8', 256, 1280, 17, 8, 8, False),
    ('B1024 8x8', 1024, 1280, 17, 8, 8, False),
    ('B32 12x12', 32, 1280, 17, 8, 12, False),
    ('B256 12x12', 256, 1280, 17, 8, 12, False),
    ('B64 16x16', 64, 1280, 17, 8, 16, False),
    ('B16 24x24', 16, 1280, 17, 8, 24, False),
    ('B64 D72', 64, 1280, 17, 72, 8, False),
    ('B32 J122 12x12', 32, 1280, 122, 8, 12, False),
    ('B64 C512', 64, 512, 17, 8, 8, False),
    ('B8 8x8', 8, 1280, 17, 8, 8, False),
    ('B320 8x8', 320, 1280, 17, 8, 8, False),
]


def timed(fn, n=20, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn()
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3


def main():
    quick = 'quick' in sys.argv
    only = os.environ.get('SWEEP_SHAPES')
    g = torch.Generator(device='cuda').manual_seed(0)
    for label, B, C, J, D, side, nhwc in SHAPES:
        if only and label not in only.split(','):
            continue
        cfg = MetrabsConfig(depth=D, proc_side=side * 32)
        feat = torch.randn(B, C, side, side, device='cuda', generator=g)
        if nhwc:
            feat = feat.contiguous(memory_format=torch.channels_last)
        w = torch.randn(J * (1 + D), C, device='cuda', generator=g) * 0.03
        bias = torch.randn(J * (1 + D), device='cuda', generator=g) * 0.1
        packed = kernels.head_pack_weights(w, bias, J, D, torch.float32)
        n_cb = -(-side * side // 64)
        out = (torch.empty(B, J, 2, device='cuda'), torch.empty(B, J, 3, device='cuda'))
        need = kernels._lib.load().mtr_head_workspace_bytes(0, 1 if nhwc else 0, B, C, side, side, J, D)
        ws = torch.empty(max(need, 8) // 8, device='cuda', dtype=torch.float64)
        conv = torch.nn.Conv2d(C, J * (1 + D), 1).cuda()
        with torch.no_grad():
            conv.weight.copy_(w[:, :, None, None])
            conv.bias.copy_(bias)
        if nhwc:
            conv = conv.to(memory_format=torch.channels_last)
        with torch.inference_mode():
            t_lib = timed(lambda: kernels.softargmax_decode(conv(feat), J, cfg))
            lib2, lib3 = kernels.softargmax_decode(conv(feat), J, cfg)
            base = kernels.head_fused(feat, packed, C, J, cfg, rt_k_groups=1, rt_loader=1, rt_split=1,
                                      rt_column_blocks=1, workspace=False)
            base = (base[0].clone(), base[1].clone())
            print(json.dumps(dict(shape=label, choice='library conv + decode', us=round(t_lib, 2),
                                  max_abs_vs_base_mm=float((lib3 - base[1]).abs().max()))), flush=True)
            grid = [dict()]  # the library's own choice (with a workspace)
            grid.append(dict(rt_loader=1, rt_split=1))  # round-2 dispatch
            atoms = 1 if 1 + D <= 16 else 0
            tiles = [0] if not atoms else ([0, 2, 3, 5] if quick else [0, 1, 2, 3, 4, 5])
            splits = [1, 2] if n_cb >= 2 else [1]
            for ld, ks, rt, sp in itertools.product([2, 1], [1, 2], tiles, splits):
                if ld == 2 and ks == 2:
                    continue
                if ld == 1 and ks == 2 and rt > 3:
                    continue
                grid.append(dict(rt_loader=ld, rt_k_groups=ks, rt_tiles=rt, rt_split=sp, rt_column_blocks=1))
            if n_cb >= 2 and atoms:
                for np_ in (2, 3, 4):
                    if np_ <= max(2, n_cb):
                        grid.append(dict(rt_loader=1, rt_split=1, rt_column_blocks=np_))
            for opts in grid:
                def call():
                    return kernels.head_fused(feat, packed, C, J, cfg, out=out, workspace=ws if need else False,
                                              **opts)
                try:
                    t = timed(call)
                    o2, o3 = call()
                    torch.cuda.synchronize()
                    rec = dict(shape=label, choice=opts or 'auto', us=round(t, 2),
                               bit_equal=bool(torch.equal(o2, base[0]) and torch.equal(o3, base[1])),
                               max_abs_vs_base_mm=float((o3 - base[1]).abs().max()))
                except Exception as e:  # noqa: BLE001
                    rec = dict(shape=label, choice=opts or 'auto', error=str(e)[:200])
                print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
