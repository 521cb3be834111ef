"""NCHW soft-argmax decode timings (tools/microbench.bench_decode) + output hashes (the same bits are expected from
every mapping of joints to waves).  Run on the GPU box."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from metrabs_amd import _lib, kernels
from metrabs_amd.config import MetrabsConfig
from tools.microbench import timeit
if len(sys.argv) > 1:
    _lib.load(sys.argv[1])
for B, J, D, side, dt in [(32768, 17, 8, 8, torch.float32), (65536, 17, 8, 8, torch.float16), (32768, 17, 8, 8, torch.bfloat16),
                          (2048, 122, 8, 12, torch.float32), (4096, 17, 72, 8, torch.float32), (8192, 24, 8, 8, torch.float32),
                          (64, 17, 8, 8, torch.float32), (8192, 17, 8, 12, torch.float16), (333, 17, 8, 8, torch.float32)]:
    cfg = MetrabsConfig(depth=D, proc_side=side * 32)
    g = torch.Generator(device='cuda').manual_seed(0)
    x = (torch.randn(B, J * (1 + D), side, side, device='cuda', generator=g) * 3).to(dt)
    t = min(timeit(lambda: kernels.softargmax_decode(x, J, cfg)) for _ in range(3))
    c2, c3 = kernels.softargmax_decode(x, J, cfg)
    nbytes = x.numel() * x.element_size() + B * J * 20
    print(json.dumps(dict(lib=os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else 'product', shape=[B, J, D, side], dtype=str(dt).split('.')[-1],
                          us=round(t * 1e6, 1), frac_of_8TBps=round(nbytes / t / 8e12, 4),
                          sha256_16=hashlib.sha256(c3.cpu().numpy().tobytes() + c2.cpu().numpy().tobytes()).hexdigest()[:16])), flush=True)
    del x
    torch.cuda.empty_cache()
