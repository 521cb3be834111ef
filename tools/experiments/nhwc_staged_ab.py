"""NHWC soft-argmax decode: the LDS-staged kernel (round 6: a crop copied into LDS with 16-byte-per-lane
global_load_lds, then the row walk out of LDS; mtr_softargmax_decode_opts nhwc_staging = 2) against the kernel that
walks global memory (nhwc_staging = 1), same process, alternated; bit-equality by hash.  One JSON line per
(variant, shape, dtype).  Run on the GPU box:  python tools/experiments/nhwc_staged_ab.py [developer .so]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SHAPES = [(32768, 17, 8, 8), (4096, 17, 8, 8), (1024, 17, 8, 8), (256, 17, 8, 8), (8192, 24, 8, 8), (16384, 5, 8, 12),
          (8192, 4, 8, 16), (8192, 17, 8, 12), (2048, 122, 8, 12)]


def main():
    import torch
    from metrabs_amd import _lib, kernels
    tag = ''
    if len(sys.argv) > 1:   # a developer build (tools/experiments/variant_lib.py), e.g. -DMTR_NHWC_RING=6
        _lib.load(sys.argv[1])
        tag = os.path.basename(sys.argv[1])
    from metrabs_amd.config import MetrabsConfig
    from tools.microbench import timeit
    for B, J, D, side in SHAPES:
        for dt in (torch.float32, torch.float16):
            cfg = MetrabsConfig(depth=D, proc_side=side * 32)
            g = torch.Generator(device='cuda').manual_seed(1)
            cl = (torch.randn(B, side, side, J * (1 + D), device='cuda', generator=g) * 3).to(dt).permute(0, 3, 1, 2)
            nbytes = cl.numel() * cl.element_size() + B * J * 20
            for rnd in range(2):
                for name, mode in (('walk_global', 1), ('staged_lds', 2), ('staged_two_crops', 3), ('library_rule', 0)):
                    t = min(timeit(lambda: kernels.softargmax_decode(cl, J, cfg, nhwc_staging=mode)) for _ in range(3))
                    c2, c3 = kernels.softargmax_decode(cl, J, cfg, nhwc_staging=mode)
                    print(json.dumps(dict(lib=tag, variant=name, shape=[B, J, D, side, side], dtype=str(dt).split('.')[-1],
                                          us=round(t * 1e6, 1), frac_of_8TBps=round(nbytes / t / 8e12, 4),
                                          sha256_16=hashlib.sha256(c3.cpu().numpy().tobytes() + c2.cpu().numpy().tobytes()).hexdigest()[:16])),
                          flush=True)
            del cl
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
