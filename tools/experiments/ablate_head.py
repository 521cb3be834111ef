"""Timing ablations of head_fused32_kernel (developer tool, not part of the library).

    python tools/experiments/ablate_head.py build      # here (hipcc cross-compiles): one .so per mask
    python tools/experiments/ablate_head.py run        # on the GPU box: times every variant

MTR_ABLATE bits (metrabs_amd/csrc/head_fused.hip): 1 = no decode epilogue, 2 = no MFMA/carry,
4 = no global loads / LDS stores inside the K loop, 8 = no f64 carry, 16 = no LDS stores (loads
still waited for), 32 = no global loads (LDS stores of stale registers).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tools', 'experiments', '_build')
MASKS = [0, 4, 16, 32]


def build():
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, 'metrabs_amd', 'csrc')
    srcs = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith('.hip'))
    procs = []
    for m in MASKS:
        cmd = ['hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950',
               f'-DMTR_ABLATE={m}', '-I', os.path.join(ROOT, 'include'), *srcs,
               '-o', os.path.join(OUT, f'libmtr_abl{m}.so')]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE))
    for p in procs:
        _, err = p.communicate()
        if p.returncode:
            sys.exit(err.decode())


def run_one(mask):
    os.environ['MTR_HEAD_W8'] = '0'  # the hooks live in the 4-wave kernel; small launches default to 8 waves
    sys.path.insert(0, ROOT)
    import torch
    from metrabs_amd import _lib
    _lib.load(os.path.join(OUT, f'libmtr_abl{mask}.so'))
    from metrabs_amd import kernels
    from metrabs_amd.config import MetrabsConfig
    g = torch.Generator(device='cuda').manual_seed(0)
    res = {'mask': mask}
    for name, B, dt, nhwc in [('B64 f32', 64, torch.float32, False), ('B64 f32 nhwc', 64, torch.float32, True),
                              ('B64 f16', 64, torch.float16, False), ('B1024 f32', 1024, torch.float32, False),
                              ('B1024 f16', 1024, torch.float16, False)]:
        feat = torch.randn(B, 1280, 8, 8, device='cuda', generator=g).to(dt)
        if nhwc:
            feat = feat.contiguous(memory_format=torch.channels_last)
        w = torch.randn(153, 1280, device='cuda', generator=g) * 0.03
        packed = kernels.head_pack_weights(w, torch.zeros(153, device='cuda'), 17, 8, dt)
        o = (torch.empty(B, 17, 2, device='cuda'), torch.empty(B, 17, 3, device='cuda'))
        cfg = MetrabsConfig()
        for _ in range(5):
            kernels.head_fused(feat, packed, 1280, 17, cfg, out=o)
        torch.cuda.synchronize()
        # 20 launches per graph replay: GPU time per launch without the Python/ctypes call floor
        n = 20
        st = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            kernels.head_fused(feat, packed, 1280, 17, cfg, out=o)
            st.synchronize()
            with torch.cuda.graph(graph, stream=st):
                for _ in range(n):
                    kernels.head_fused(feat, packed, 1280, 17, cfg, out=o)
        graph.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        a.record()
        for _ in range(reps):
            graph.replay()
        b.record()
        torch.cuda.synchronize()
        res[name] = round(a.elapsed_time(b) / (n * reps) * 1e3, 1)
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    elif sys.argv[1] == 'run':
        for m in MASKS:
            subprocess.run([sys.executable, __file__, 'one', str(m)])
    else:
        run_one(int(sys.argv[2]))
