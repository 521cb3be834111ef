"""mtr_softargmax_decode on NHWC logits (the TF twin's layout) beside NCHW, graph-replayed (developer tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402
from tools.microbench import timeit  # noqa: E402

for B, J, D, side in [(64, 17, 8, 8), (8, 17, 8, 8), (256, 17, 8, 8), (1024, 17, 8, 8), (4096, 17, 8, 8), (32, 122, 8, 12),
                      (64, 17, 72, 8), (32768, 17, 8, 8), (2048, 122, 8, 12)]:
    cfg = MetrabsConfig(depth=D, proc_side=side * 32)
    g = torch.Generator(device='cuda').manual_seed(1)
    lg = torch.randn(B, J * (1 + D), side, side, device='cuda', generator=g)
    cl = lg.contiguous(memory_format=torch.channels_last)
    t1 = timeit(lambda: kernels.softargmax_decode(lg, J, cfg))
    t2 = timeit(lambda: kernels.softargmax_decode(cl, J, cfg))
    a, b = kernels.softargmax_decode(lg, J, cfg), kernels.softargmax_decode(cl, J, cfg)
    nbytes = lg.numel() * 4 + B * J * 20
    print(f'B={B} J={J} D={D} {side}x{side}: NCHW {t1 * 1e6:.1f} us ({nbytes / t1 / 8e12:.2f} of 8 TB/s), NHWC {t2 * 1e6:.1f} us '
          f'({nbytes / t2 / 8e12:.2f}), max |diff| {float((a[1] - b[1]).abs().max()):.1e} mm')
