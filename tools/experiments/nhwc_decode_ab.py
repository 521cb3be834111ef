"""NHWC soft-argmax decode (mtr_softargmax_decode on channels_last logits, csrc/decode.hip): the round-6 walk -- the
next batch of loads requested before the current one is summed -- against the round-5 walk (-DMTR_NHWC_PREFETCH=0),
same launches, alternated processes, bit-equality by hash.  `build` here (seconds: decode.hip + the product's objects),
`run` on the GPU box.  One JSON line per (variant, shape, dtype)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tools', 'experiments', '_build')
VARIANTS = {'per_walk_kernels': [], 'one_kernel_r05': ['-DMTR_NHWC_ONE_KERNEL=1'], 'prefetch': ['-DMTR_NHWC_PREFETCH=1']}
SHAPES = [(32768, 17, 8, 8), (4096, 17, 8, 8), (256, 17, 8, 8), (2048, 122, 8, 12), (8192, 17, 8, 12), (4096, 17, 8, 16)]


def build():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    from metrabs_amd import build as product
    product.build_library(verbose=False)
    others = [os.path.join(product.BUILD_DIR, f + '.o') for f in product.sources() if f != 'decode.hip']
    for name, defs in VARIANTS.items():
        obj = os.path.join(OUT, f'decode_{name}.o')
        subprocess.run(['hipcc', *product.FLAGS, *defs, '-c', os.path.join(product.CSRC, 'decode.hip'), '-o', obj], check=True,
                       stderr=subprocess.DEVNULL)
        subprocess.run(['hipcc', '-shared', '-fPIC', f'--offload-arch={product.ARCH}', obj, *others, '-o',
                        os.path.join(OUT, f'libmtr_decode_{name}.so')], check=True)
        os.remove(obj)


def run_one(name):
    sys.path.insert(0, ROOT)
    import torch
    from metrabs_amd import _lib
    _lib.load(os.path.join(OUT, f'libmtr_decode_{name}.so'))
    from metrabs_amd import kernels
    from metrabs_amd.config import MetrabsConfig
    from tools.microbench import timeit
    for B, J, D, side in SHAPES:
        for dt in (torch.float32, torch.float16):
            cfg = MetrabsConfig(depth=D, proc_side=side * 32)
            g = torch.Generator(device='cuda').manual_seed(1)
            lg = (torch.randn(B, J * (1 + D), side, side, device='cuda', generator=g) * 3).to(dt)
            cl = lg.contiguous(memory_format=torch.channels_last)
            del lg
            t = min(timeit(lambda: kernels.softargmax_decode(cl, J, cfg)) for _ in range(3))
            c2, c3 = kernels.softargmax_decode(cl, J, cfg)
            nbytes = cl.numel() * cl.element_size() + B * J * 20
            print(json.dumps(dict(variant=name, shape=[B, J, D, side, side], dtype=str(dt).split('.')[-1], us=round(t * 1e6, 1),
                                  frac_of_8TBps=round(nbytes / t / 8e12, 4),
                                  sha256_16=hashlib.sha256(c3.cpu().numpy().tobytes() + c2.cpu().numpy().tobytes()).hexdigest()[:16])),
                  flush=True)
            del cl
            torch.cuda.empty_cache()


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    elif sys.argv[1] == 'run':
        for rnd in range(2):
            for name in VARIANTS:
                subprocess.run([sys.executable, __file__, 'one', name])
    else:
        run_one(sys.argv[2])
