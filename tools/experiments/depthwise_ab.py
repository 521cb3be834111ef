"""K11 (depthwise 3x3 + bias + SiLU + plane mean in one pass, csrc/depthwise.hip) on the stride-1 layers of EfficientNetV2-S at 256 px,
six input tensors in rotation (not cache-resident), per developer build (variant_lib.py).  Round 6: two-row instead of four-row
output blocks per lane (62 instead of 78 registers: 8 instead of 6 waves per SIMD; a source edit that was not kept): the same or
slower (profiles/r06zh_depthwise_rows.jsonl: 960 ch 16x16 f32 25.7 vs 25.6 us, 512 ch 11.7 -> 13.2, f16 16.7 -> 19.4).
    python tools/experiments/depthwise_ab.py [lib.so ...]      (on the GPU box)"""
import hashlib
import json
import os
import subprocess
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
    for shape, dt in [((64, 512, 16, 16), torch.float32), ((64, 960, 16, 16), torch.float32), ((64, 1536, 8, 8), torch.float32),
                      ((64, 960, 16, 16), torch.float16), ((64, 1536, 8, 8), torch.float16), ((64, 256, 32, 32), torch.float32)]:
        xs = [torch.randn(shape, device='cuda', generator=g).to(dt) for _ in range(6)]
        w = torch.randn(shape[1], 3, 3, device='cuda', generator=g) * 0.3
        b = torch.randn(shape[1], device='cuda', generator=g) * 0.1
        k = [0]

        def step():
            k[0] += 1
            return kernels.depthwise3x3_bias_act(xs[k[0] % 6], w, b, 'silu', 1, 1, want_mean=True)
        t = min(timeit(step, iters=30) for _ in range(3))
        y, m = kernels.depthwise3x3_bias_act(xs[0], w, b, 'silu', 1, 1, want_mean=True)
        nbytes = 2 * y.numel() * y.element_size()
        print(json.dumps(dict(lib=os.path.basename(lib) if lib else 'product', shape=list(shape), dtype=str(dt).split('.')[-1],
                              us=round(t * 1e6, 2), frac_of_8TBps=round(nbytes / t / 8e12, 4),
                              sha=hashlib.sha1(y.float().cpu().numpy().tobytes() + m.cpu().numpy().tobytes()).hexdigest()[:12])), flush=True)
        del xs


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'one':
        run(sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != '-' else None)
    else:
        for _ in range(2):
            for lib in (sys.argv[1:] or ['-']):
                subprocess.run([sys.executable, __file__, 'one', lib])
