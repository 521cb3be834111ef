"""Fused head (hand-written MFMA + decode epilogue) vs the unfused pair (library 1x1 conv + HIP
decode kernel) at bench shapes, graph-replayed.  Developer probe."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd.config import MetrabsConfig
from metrabs_amd.models.metrabs import MetrabsHeads
from oracle import cases, cpu_ref

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

for (B, C, J, D, H, dt) in [(64, 1280, 17, 8, 8, torch.float32), (1024, 1280, 17, 8, 8, torch.float32),
                            (64, 1280, 17, 8, 8, torch.float16), (1024, 1280, 17, 8, 8, torch.float16),
                            (64, 1280, 17, 8, 8, torch.bfloat16), (32, 1280, 122, 8, 12, torch.float16),
                            (256, 1280, 122, 8, 12, torch.float16), (256, 2048, 24, 8, 8, torch.float16),
                            (64, 1280, 17, 72, 8, torch.float32), (64, 1280, 17, 72, 8, torch.float16),
                            (1024, 1280, 17, 72, 8, torch.float16), (16, 1280, 17, 8, 24, torch.float16),
                            (64, 1280, 17, 8, 20, torch.bfloat16)]:
    cfg = MetrabsConfig(depth=D, proc_side=H * 32)
    heads = MetrabsHeads(J, cfg, in_channels=C, fused=True).cuda()
    feat = torch.randn(B, C, H, H, device='cuda').to(dt)
    w = heads.conv_final.weight.detach().float().cpu()[:, :, 0, 0]; b = heads.conv_final.bias.detach().float().cpu()
    res = {}
    with torch.inference_mode():
        for fused in (True, False):
            heads.fused = fused
            if dt != torch.float32:
                f = lambda: heads(feat) if fused else torch.autocast('cuda', dtype=dt).__enter__() or heads(feat)
            run = (lambda: heads(feat))
            if not fused and dt != torch.float32:
                hh = heads.half() if dt == torch.float16 else heads.bfloat16()
                run = (lambda: hh(feat))
            try:
                res[fused] = timed(run)
                out = run()
            except Exception as e:
                res[fused] = float('nan'); print('error', fused, str(e)[:100])
            if dt != torch.float32:
                heads = heads.float()
        o2, o3 = cpu_ref.heads_forward(feat[:4].float().cpu(), w, b, J, cpu_ref.HeadConfig(depth=D, proc_side=H * 32))
    print(f'B={B} C={C} J={J} D={D} {H}x{H} {dt}: fused {res[True]:.1f} us, unfused {res[False]:.1f} us')
