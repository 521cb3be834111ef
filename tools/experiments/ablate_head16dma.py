"""Timing ablations of head_fused16dma_kernel (developer tool; MTR_H16_DMA_ABLATE bits of head_fused.hip).

    python tools/experiments/ablate_head16dma.py build   # here: one .so per variant (hipcc, no GPU)
    python tools/experiments/ablate_head16dma.py run     # on the GPU box: one JSON line per (variant, case)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tools', 'experiments', '_build')
# (bit 2 also lets the compiler drop the MFMAs whose results are no longer stored: read "nostore" rows as
#  "no epilogue and a third of the MFMAs")
VARIANTS = {'base': 0, 'nodecode': 1, 'nostore_nodecode': 2, 'nomfma': 4, 'nocopies': 8, 'noreads': 16,
            'copies_barriers_only': 2 | 4 | 16, 'nothing': 2 | 4 | 8 | 16}
EARLY = os.environ.get('H16_EARLY', '0')
CASES = [(256, 122, 12, False), (256, 122, 12, True), (32, 122, 12, False), (64, 17, 8, False), (1024, 17, 8, False)]


def build():
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, 'metrabs_amd', 'csrc')
    procs = []
    for name, bits in VARIANTS.items():
        obj = os.path.join(OUT, f'h16dma{EARLY}_{name}.o')
        cmd = ['hipcc', '-O3', '-std=c++17', '-fPIC', '-c', '--offload-arch=gfx950', f'-DMTR_H16_DMA_ABLATE={bits}', f'-DMTR_H16_EARLY_DEFAULT={EARLY}',
               '-I', os.path.join(ROOT, 'include'), os.path.join(csrc, 'head_fused.hip'), '-o', obj]
        procs.append((name, obj, subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
    for name, obj, p in procs:
        _, err = p.communicate()
        if p.returncode:
            sys.exit(err.decode()[-3000:])
        # (+ the row-tile core the entry points link against, from the library's own build directory)
        r = subprocess.run(['hipcc', '-shared', '-fPIC', '--offload-arch=gfx950', obj,
                            os.path.join(csrc, 'build', 'head_rt.hip.o'), '-o',
                            os.path.join(OUT, f'libmtr_h16dma{EARLY}_{name}.so')], capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])


def run():
    sys.path.insert(0, ROOT)
    import ctypes
    import torch
    from bench import graph_time
    from metrabs_amd import _lib, kernels
    from metrabs_amd.config import MetrabsConfig
    base = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(0)
    for name in VARIANTS:
        var = ctypes.CDLL(os.path.join(OUT, f'libmtr_h16dma{EARLY}_{name}.so'))
        var.mtr_head_fused.restype = ctypes.c_int
        var.mtr_head_fused.argtypes = base.mtr_head_fused.argtypes
        for B, J, side, nhwc in CASES:
            cfg = MetrabsConfig(depth=8, proc_side=side * 32)
            feat = torch.randn(B, 1280, side, side, device='cuda', generator=g).half()
            if nhwc:
                feat = feat.contiguous(memory_format=torch.channels_last)
            w = torch.randn(J * 9, 1280, device='cuda', generator=g) * 0.03
            packed = kernels.head_pack_weights(w, torch.zeros(J * 9, device='cuda'), J, 8, torch.float16)
            c2 = torch.empty(B, J, 2, device='cuda')
            c3 = torch.empty(B, J, 3, device='cuda')
            hp = cfg.head_params()

            def call():
                rc = var.mtr_head_fused(ctypes.c_void_p(feat.data_ptr()), _lib.MTR_F16, 1 if nhwc else 0, B, 1280, side,
                                        side, ctypes.c_void_p(packed.data_ptr()), J, 8, ctypes.byref(hp),
                                        ctypes.c_void_p(c2.data_ptr()), ctypes.c_void_p(c3.data_ptr()),
                                        _lib.current_stream_ptr(feat.device))
                assert rc == 0, rc
            call()
            us = graph_time([call] * 20, 5) * 1e6
            print(json.dumps(dict(early=EARLY, variant=name, bits=VARIANTS[name], B=B, J=J, side=side, nhwc=nhwc, us=round(us, 2))),
                  flush=True)


if __name__ == '__main__':
    with __import__('contextlib').nullcontext():
        (build if sys.argv[1:] == ['build'] else run)()
