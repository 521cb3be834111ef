"""NHWC 16-bit fused head, graph-replayed timing for explicit dispatch choices (HEAD_DMA = 0 / 1,
HEAD_GPW = 1..3 in the environment of THIS tool -> mtr_head_options): developer A/B probe."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd import kernels
from metrabs_amd.config import MetrabsConfig


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


res = {}
gen = torch.Generator(device='cuda').manual_seed(0)
for name, B, C, J, side in [('B64', 64, 1280, 17, 8), ('B256', 256, 1280, 17, 8), ('B1024', 1024, 1280, 17, 8),
                            ('B4096', 4096, 1280, 17, 8), ('J122 B32', 32, 1280, 122, 12), ('J122 B256', 256, 1280, 122, 12),
                            ('C2048 J24 B256', 256, 2048, 24, 8)]:
    feat = torch.randn(B, C, side, side, device='cuda', generator=gen).half().contiguous(memory_format=torch.channels_last)
    w = torch.randn(J * 9, C, device='cuda', generator=gen) * 0.03
    packed = kernels.head_pack_weights(w, torch.zeros(J * 9, device='cuda'), J, 8, torch.float16)
    cfg = MetrabsConfig(proc_side=side * 32)
    o = (torch.empty(B, J, 2, device='cuda'), torch.empty(B, J, 3, device='cuda'))
    res[name] = round(timed(lambda: kernels.head_fused(feat, packed, C, J, cfg, out=o, dma_staging=int(os.environ.get('HEAD_DMA', '-1')), groups_per_workgroup=int(os.environ.get('HEAD_GPW', '0')))), 1)
print(f"DMA={os.environ.get('HEAD_DMA', 'auto')} GPW={os.environ.get('HEAD_GPW', 'auto')}: {res}", flush=True)
