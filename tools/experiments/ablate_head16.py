"""Timing ablations / build-knob sweeps of head_fused16_kernel (developer tool).

    python tools/experiments/ablate_head16.py build   # here: one .so per variant
    python tools/experiments/ablate_head16.py run     # on the GPU box

Variants = -D flags of metrabs_amd/csrc/head_fused.hip: MTR_H16_ABLATE bits (1 no global loads in
the K loop, 2 no MFMA, 4 no LDS stores, 8 no LDS reads), MTR_H16_AHEAD, MTR_H16_MINWAVES."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tools', 'experiments', '_build')
ABLATIONS = {'base': [], 'noload': ['-DMTR_H16_ABLATE=1'], 'nomfma': ['-DMTR_H16_ABLATE=2'],
             'nostore': ['-DMTR_H16_ABLATE=4'], 'noread': ['-DMTR_H16_ABLATE=8'],
             'noload_nostore': ['-DMTR_H16_ABLATE=5'], 'onlyloads': ['-DMTR_H16_ABLATE=14']}
KNOBS = {f'a{a}w{w}': [f'-DMTR_H16_AHEAD={a}', f'-DMTR_H16_MINWAVES={w}']
         for a, w in [(1, 1), (2, 1)]}
VARIANTS = KNOBS if os.environ.get('H16_SWEEP') == 'knobs' else ABLATIONS
GPWS = ['1', '2', '3'] if os.environ.get('H16_SWEEP') == 'knobs' else ['0']
CASES = [('J122 12x12 B256 nchw', 256, 122, 12, False), ('J122 12x12 B256 nhwc', 256, 122, 12, True),
         ('J17 8x8 B1024 nchw', 1024, 17, 8, False), ('J17 8x8 B64 nchw', 64, 17, 8, False)]


def build():
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, 'metrabs_amd', 'csrc')
    srcs = [os.path.join(csrc, 'head_fused.hip')]
    procs = []
    for name, flags in VARIANTS.items():
        cmd = ['hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', *flags,
               '-I', os.path.join(ROOT, 'include'), *srcs, '-o', os.path.join(OUT, f'libmtr_h16_{name}.so')]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE))
    for p in procs:
        _, err = p.communicate()
        if p.returncode:
            sys.exit(err.decode())


def run_one(name):
    sys.path.insert(0, ROOT)
    import ctypes
    import torch
    from metrabs_amd import _lib
    _lib.load()  # the full library first (head_fused.hip alone lacks the other entry points)
    from metrabs_amd import kernels
    from metrabs_amd.config import MetrabsConfig
    var = ctypes.CDLL(os.path.join(OUT, f'libmtr_h16_{name}.so'))
    base = _lib.load()
    var.mtr_head_fused.restype = ctypes.c_int
    var.mtr_head_fused.argtypes = base.mtr_head_fused.argtypes
    g = torch.Generator(device='cuda').manual_seed(0)
    res = {'variant': name, 'gpw': os.environ.get('HEAD_GPW', '0')}
    for cname, B, J, side, nhwc in CASES:
        feat = torch.randn(B, 1280, side, side, device='cuda', generator=g).half()
        if nhwc:
            feat = feat.contiguous(memory_format=torch.channels_last)
        w = torch.randn(J * 9, 1280, device='cuda', generator=g) * 0.03
        packed = kernels.head_pack_weights(w, torch.zeros(J * 9, device='cuda'), J, 8, torch.float16)
        c2 = torch.empty(B, J, 2, device='cuda'); c3 = torch.empty(B, J, 3, device='cuda')
        hp = MetrabsConfig(proc_side=side * 32).head_params()
        st = torch.cuda.Stream()
        def call():
            rc = var.mtr_head_fused(feat.data_ptr(), _lib.MTR_F16, _lib.MTR_NHWC if nhwc else _lib.MTR_NCHW,
                                    B, 1280, side, side, packed.data_ptr(), J, 8, ctypes.byref(hp),
                                    c2.data_ptr(), c3.data_ptr(), st.cuda_stream)
            assert rc == 0, rc
        with torch.cuda.stream(st):
            for _ in range(3):
                call()
            st.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(10):
                    call()
        gr.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            gr.replay()
        b.record(); torch.cuda.synchronize()
        res[cname] = round(a.elapsed_time(b) / 50 * 1e3, 1)
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    elif sys.argv[1] == 'run':
        for name in VARIANTS:
            for gpw in GPWS:
                subprocess.run([sys.executable, os.path.abspath(__file__), 'one', name], check=False,
                               env=dict(os.environ, HEAD_GPW=gpw))
    else:
        run_one(sys.argv[2])
