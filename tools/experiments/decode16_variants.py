"""A/B of the 16-bit heads' decode epilogue (csrc/head16.h) -- compile-time variants of the three sources that
include it, built HERE, timed on the GPU box.  Output hash: the SAME bits are expected from every variant.
    python tools/experiments/decode16_variants.py build | run"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tools', 'experiments', '_build')
SRCS = ['head_fused.hip', 'head_areg.hip', 'head_res.hip']
VARIANTS = {'loop': ['-DMTR_DECODE_DC8=0'], 'dc8': ['-DMTR_DECODE_DC8=1']}
# (round 6: a third variant, f32 depth sums per position -- -DMTR_DECODE_COL32, a source edit that was not kept -- measured
#  with this script: profiles/r06t_decode16_col32.jsonl)


def build():
    sys.path.insert(0, ROOT)
    from metrabs_amd import build as b
    b.build_library(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    others = [os.path.join(b.BUILD_DIR, f + '.o') for f in b.sources() if f not in SRCS]
    for name, defs in VARIANTS.items():
        objs, procs = [], []
        for src in SRCS:
            obj = os.path.join(OUT, f'{name}_{src}.o')
            objs.append(obj)
            procs.append(subprocess.Popen([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(src, []), *defs, '-c',
                                           os.path.join(b.CSRC, src), '-o', obj], stderr=subprocess.DEVNULL))
        for p in procs:
            if p.wait():
                sys.exit(f'{name}: compile failed')
        subprocess.run([b._hipcc(), '-shared', '-fPIC', f'--offload-arch={b.ARCH}', *others, *objs, '-o',
                        os.path.join(OUT, f'libmtr_dec_{name}.so')], check=True)
    print('built', len(VARIANTS))


def run_one(name):
    sys.path.insert(0, ROOT)
    import torch
    from metrabs_amd import _lib
    _lib.load(os.path.join(OUT, f'libmtr_dec_{name}.so'))
    from bench import graph_time
    from metrabs_amd import kernels
    from metrabs_amd.config import MetrabsConfig
    g = torch.Generator(device='cuda').manual_seed(3)
    with torch.inference_mode():
        for B, C, J, D, H, W in [(32, 1280, 122, 8, 12, 12), (256, 1280, 122, 8, 12, 12), (64, 1280, 17, 8, 8, 8),
                                 (1024, 1280, 17, 8, 8, 8), (320, 1280, 17, 8, 8, 8), (64, 1280, 17, 8, 16, 16),
                                 (64, 1280, 17, 4, 8, 8)]:
            for nhwc in (False, True):
                cfg = MetrabsConfig(depth=D, proc_side=max(H, W) * 32)
                feat = torch.randn(B, C, H, W, device='cuda', generator=g).half()
                if nhwc:
                    feat = feat.contiguous(memory_format=torch.channels_last)
                w = torch.randn(J * (1 + D), C, device='cuda', generator=g) * 0.02
                b = torch.randn(J * (1 + D), device='cuda', generator=g) * 0.1
                packed = kernels.head_pack_weights(w, b, J, D, torch.float16)
                out = kernels.head_fused(feat, packed, C, J, cfg)
                h = hashlib.sha1(out[0].cpu().numpy().tobytes() + out[1].cpu().numpy().tobytes()).hexdigest()[:12]
                us = graph_time([lambda: kernels.head_fused(feat, packed, C, J, cfg)] * 10, 5) * 1e6
                print(json.dumps(dict(variant=name, shape=[B, C, J, D, H, W], nhwc=nhwc, us=round(us, 2), sha=h)),
                      flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    elif sys.argv[1] == 'run':
        for m in list(VARIANTS) * 2:    # (each variant twice, alternating: the box's clock drifts)
            subprocess.run([sys.executable, __file__, 'one', m])
    else:
        run_one(sys.argv[2])
