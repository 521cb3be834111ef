"""K10 (bias + activation in place, csrc/bias_act.hip) on activations that are NOT cache-resident: four tensors of the shape
in rotation (> the 256 MiB Infinity Cache for the large shapes), per developer build (variant_lib.py).  Round 6: 2 / 4 / 8 vectors per
thread with their loads issued together (a source edit that was not kept) against the shipped one vector per thread and up to 32
workgroups per CU -- the shipped launch is the fastest everywhere (201 MB in place: 36.1 us = 0.70 of 8 TB/s cold; 39.7 / 44.6 / 62.9
at 2 / 4 / 8: profiles/r06za_bias_act_ab.jsonl).
    python tools/experiments/bias_act_ab.py [lib.so ...]      (on the GPU box)"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(lib):
    import torch
    from metrabs_amd import _lib
    if lib:
        _lib.load(lib)
    from metrabs_amd import kernels
    from tools.microbench import timeit
    g = torch.Generator(device='cuda').manual_seed(0)
    for shape, dt, res in [((64, 24, 128, 128), torch.float32, False), ((64, 96, 64, 64), torch.float32, False),
                           ((64, 48, 64, 64), torch.float32, True), ((64, 256, 16, 16), torch.float32, False),
                           ((64, 1536, 8, 8), torch.float32, False), ((64, 96, 64, 64), torch.float16, False),
                           ((64, 24, 128, 128), torch.float16, True)]:
        ys = [torch.randn(shape, device='cuda', generator=g).to(dt) for _ in range(4)]
        rs = [torch.randn(shape, device='cuda', generator=g).to(dt) for _ in range(4)] if res else [None] * 4
        b = torch.randn(shape[1], device='cuda', generator=g)
        k = [0]

        def step():
            j = k[0] % 4
            k[0] += 1
            kernels.bias_act_(ys[j], b, 'silu', residual=rs[j]) if res else kernels.bias_act_(ys[j], b, 'silu')
        t = min(timeit(step, iters=40) for _ in range(3))
        y0 = torch.randn(shape, device='cuda', generator=g).to(dt)
        r0 = torch.randn(shape, device='cuda', generator=g).to(dt) if res else None
        kernels.bias_act_(y0, b, 'silu', residual=r0) if res else kernels.bias_act_(y0, b, 'silu')
        nbytes = (3 if res else 2) * y0.numel() * y0.element_size()
        print(json.dumps(dict(lib=os.path.basename(lib) if lib else 'product', shape=list(shape), dtype=str(dt).split('.')[-1],
                              residual=res, us=round(t * 1e6, 2), frac_of_8TBps=round(nbytes / t / 8e12, 4),
                              sha=hashlib.sha1(y0.float().cpu().numpy().tobytes()).hexdigest()[:12])), flush=True)
        del ys, rs


if __name__ == '__main__':
    import subprocess
    if len(sys.argv) > 1 and sys.argv[1] == 'one':
        run(sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != '-' else None)
    else:
        libs = sys.argv[1:] or ['-']
        for _ in range(2):
            for lib in libs:
                subprocess.run([sys.executable, __file__, 'one', lib])
