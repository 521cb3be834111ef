"""Which block shape wins on small launches of large maps (developer tool): times mtr_head_fused_opts
over rt_tiles x rt_column_blocks x rt_k_groups for a few (B, H) shapes, graph-replayed."""
import itertools
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd import kernels
from metrabs_amd.config import MetrabsConfig


def timeit(fn, n=20, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn()
        st.synchronize()
        with torch.cuda.graph(graph, stream=st):
            for _ in range(n):
                fn()
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3


def main():
    C, J, D = 1280, 17, 8
    g = torch.Generator(device='cuda').manual_seed(0)
    w = torch.randn(J * (1 + D), C, device='cuda', generator=g) * 0.03
    b = torch.zeros(J * (1 + D), device='cuda')
    packed = kernels.head_pack_weights(w, b, J, D)
    shapes = [(32, 12), (16, 24), (64, 16), (8, 12), (64, 8)]
    if os.environ.get('SMALL_SHAPES'):
        shapes = [tuple(int(x) for x in t.split('x')) for t in os.environ['SMALL_SHAPES'].split(',')]
    for B, H in shapes:
        cfg = MetrabsConfig(proc_side=H * 32, stride_test=32)
        feat = torch.randn(B, C, H, H, device='cuda', generator=g)
        out = (torch.empty(B, J, 2, device='cuda'), torch.empty(B, J, 3, device='cuda'))
        res = {}
        for rt, np_, ks in itertools.product((0, 1, 2, 3, 5), (0, 1, 2, 3), (0, 1, 2)):
            if H == 8 and np_ > 1:
                continue
            try:
                t = timeit(lambda: kernels.head_fused(feat, packed, C, J, cfg, out=out, rt_tiles=rt,
                                                      rt_column_blocks=np_, rt_k_groups=ks))
            except RuntimeError as e:
                t = None
            res[f'rt{rt}_np{np_}_ks{ks}'] = None if t is None else round(t, 1)
        best = min((v, k) for k, v in res.items() if v is not None)
        print(json.dumps(dict(B=B, H=H, default=res['rt0_np0_ks0'], best=best, all=res)), flush=True)


if __name__ == '__main__':
    main()
