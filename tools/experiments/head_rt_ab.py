"""f32 fused head: row-tile core vs the library pair (1x1 conv + HIP decode), graph-replayed, with
the distance of each to an fp64 evaluation (restated inline: tools never import oracle/).

    python tools/experiments/head_rt_ab.py [tag]
"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd.config import MetrabsConfig
from metrabs_amd.models.metrabs import MetrabsHeads


def coords3d_fp64(logits, J, D, box_mm=2200.0):
    """models/metrabs.py:78-83 in float64: joint softmax over (d, h, w), expectation per axis,
    heatmap_to_metric (models/util.py:20-33) at stride 32, centered stride."""
    B, _, H, W = logits.shape
    x = logits[:, J:].reshape(B, D, J, H, W).double()
    p = torch.softmax(x.permute(0, 2, 1, 3, 4).reshape(B, J, -1), dim=-1).reshape(B, J, D, H, W)
    lin = lambda n: torch.linspace(0, 1, n, dtype=torch.float64) if n > 1 else torch.full((1,), 0.5, dtype=torch.float64)
    cx = (p.sum((2, 3)) * lin(W)).sum(-1)
    cy = (p.sum((2, 4)) * lin(H)).sum(-1)
    cz = (p.sum((3, 4)) * lin(D)).sum(-1)
    P = H * 32
    px = lambda c: c * (P - 1 - ((P - 1) % 32)) + 16
    return torch.stack([px(cx) * box_mm / P, px(cy) * box_mm / P, cz * box_mm], dim=-1)



def timed(fn, n=20, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn(); st.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3


SHAPES = [(64, 1280, 17, 8, 8, False), (64, 1280, 17, 8, 8, True), (256, 1280, 17, 8, 8, False),
          (1024, 1280, 17, 8, 8, False), (1024, 1280, 17, 8, 8, True),
          (32, 1280, 17, 8, 12, False), (256, 1280, 17, 8, 12, False), (64, 1280, 17, 8, 16, False),
          (64, 1280, 17, 72, 8, False), (1024, 1280, 17, 72, 8, False), (32, 1280, 122, 8, 12, False),
          (64, 512, 17, 8, 8, False), (16, 1280, 17, 8, 24, False)]
tag = sys.argv[1] if len(sys.argv) > 1 else 'default'
# HEAD_NP=n in the environment of THIS tool: mtr_head_options.rt_column_blocks for the fused path
NP = int(os.environ.get('HEAD_NP', '0'))
if NP:
    from metrabs_amd import kernels as _k
    _orig = _k.head_fused
    _k.head_fused = lambda *a, **k: _orig(*a, rt_column_blocks=NP, **k)
only_fused = os.environ.get('HEAD_AB_ONLY_FUSED') == '1'
for (B, C, J, D, H, nhwc) in SHAPES:
    cfg = MetrabsConfig(depth=D, proc_side=H * 32)
    torch.manual_seed(B + C + J + D + H)
    heads = MetrabsHeads(J, cfg, in_channels=C, fused=True).cuda()
    feat = torch.randn(B, C, H, H, device='cuda')
    if nhwc:
        feat = feat.contiguous(memory_format=torch.channels_last)
    w = heads.conv_final.weight.detach().double().cpu()[:, :, 0, 0]
    b = heads.conv_final.bias.detach().double().cpu()
    row = dict(tag=tag, B=B, C=C, J=J, D=D, hw=H * H, nhwc=nhwc)
    with torch.inference_mode():
        sub = feat[:4].double().cpu().contiguous()
        logits = torch.einsum('nc,bchw->bnhw', w, sub) + b[None, :, None, None]
        t3 = coords3d_fp64(logits, J, D)
        for fused in ((True,) if only_fused else (True, False)):
            heads.fused = fused
            key = 'fused' if fused else 'library'
            try:
                row[key + '_us'] = round(timed(lambda: heads(feat)), 2)
                c2, c3 = heads(feat)
                row[key + '_max_vs_fp64_mm'] = float((c3[:4].cpu().double() - t3).abs().max())
            except Exception as e:  # noqa: BLE001
                row[key + '_error'] = str(e)[:120]
    flops = 2.0 * C * J * (1 + D) * H * H * B
    if 'fused_us' in row:
        row['fused_TF'] = round(flops / row['fused_us'] / 1e6, 1)
        row['fused_frac_f32_mfma'] = round(flops / row['fused_us'] / 1e6 / 157.3, 3)
    print(json.dumps(row), flush=True)
