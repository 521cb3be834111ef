"""Launch the two NHWC soft-argmax decode kernels (walking global memory / LDS-staged) a few times on the 1.28 GB shape
(for rocprofv3 --pmc passes; tools/experiments/pmc_nhwc.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd import kernels
from metrabs_amd.config import MetrabsConfig
B, J, D, side = 32768, 17, 8, 8
cfg = MetrabsConfig(depth=D, proc_side=side * 32)
for dt in (torch.float32, torch.float16):
    g = torch.Generator(device='cuda').manual_seed(1)
    cl = (torch.randn(B, side, side, J * (1 + D), device='cuda', generator=g) * 3).to(dt).permute(0, 3, 1, 2)
    for mode in (1, 2):
        for _ in range(4):
            kernels.softargmax_decode(cl, J, cfg, nhwc_staging=mode)
    torch.cuda.synchronize()
    del cl
