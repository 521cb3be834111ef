"""Timing ablations of head_rt_kernel (developer tool, not part of the library).

    python tools/experiments/ablate_rt.py build      # here (hipcc cross-compiles): one .so per variant
    python tools/experiments/ablate_rt.py run        # on the GPU box: times every variant

MTR_RT_ABLATE bits (metrabs_amd/csrc/head_rt.hip): 1 = no decode epilogue, 2 = no MFMA, 8 = no f64
carry, 16 = no fragment reads (27 = the copies + barriers alone).  MTR_RT_NBUF / MTR_RT_KS_NBUF = ring depth.
Probes that were built on top of this tool in round 2 and removed again after their result went into
DESIGN.md: K order rotated per workgroup, non-temporal copies, s_sleep stagger of the second K group,
two K groups at every launch size, hand-over cost split (no adds / no LDS traffic / no carries).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tools', 'experiments', '_build')
VARIANTS = {'full': [], 'decode_rows': ['-DMTR_RT_DECODE_PER_ROW=1'],
            'decode_rows_nodecode': ['-DMTR_RT_DECODE_PER_ROW=1', '-DMTR_RT_ABLATE=1'], 'exp32': ['-DMTR_RT_EXP32=1'], 'nodecode': ['-DMTR_RT_ABLATE=1'], 'nomfma': ['-DMTR_RT_ABLATE=2'],
            'nocarry': ['-DMTR_RT_ABLATE=8'],
            'nofrag': ['-DMTR_RT_ABLATE=16'],
            'nbuf2': ['-DMTR_RT_NBUF=2'], 'nbuf4': ['-DMTR_RT_NBUF=4'], 'nbuf8': ['-DMTR_RT_NBUF=8'], 'a18': ['-DMTR_RT_ABLATE=18'], 'a26': ['-DMTR_RT_ABLATE=26'], 'a27': ['-DMTR_RT_ABLATE=27'],
            'ks4': ['-DMTR_RT_KS_NBUF=4'], 'pair': ['-DMTR_RT_PAIR=1'],
            'nocopy': ['-DMTR_RT_ABLATE=4'], 'a5': ['-DMTR_RT_ABLATE=5'], 'a7': ['-DMTR_RT_ABLATE=7'],
            'a3': ['-DMTR_RT_ABLATE=3'], 'a21': ['-DMTR_RT_ABLATE=21'], 'a23': ['-DMTR_RT_ABLATE=23'],
            'x3': ['-DMTR_RT_X3=1'], 'x3_nodecode': ['-DMTR_RT_X3=1', '-DMTR_RT_ABLATE=1'],
            'sadd': ['-DMTR_RT_SCALAR_ADD=1'], 'spread': ['-DMTR_RT_SPREAD_READS=1'],
            'spread_nodecode': ['-DMTR_RT_SPREAD_READS=1', '-DMTR_RT_ABLATE=1'],
            'ld8': ['-DMTR_RT_LD_NBUF=8', '-DMTR_RT_LD_LA=4'], 'ld8la5': ['-DMTR_RT_LD_NBUF=8', '-DMTR_RT_LD_LA=5'],
            'ld4la2': ['-DMTR_RT_LD_NBUF=4', '-DMTR_RT_LD_LA=2'], 'ld2': ['-DMTR_RT_LD_NBUF=2', '-DMTR_RT_LD_LA=1']}
if os.environ.get('RT_VARIANTS'):
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in os.environ['RT_VARIANTS'].split(',')}


def build():
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, 'metrabs_amd', 'csrc')
    procs = []
    for name, flags in VARIANTS.items():
        # only the two head sources + capi: the other kernels are not called here
        srcs = [os.path.join(csrc, f) for f in ('head_rt.hip', 'head_fused.hip', 'capi.hip', 'decode.hip')]
        cmd = ['hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950',
               '-mllvm', '-amdgpu-mfma-vgpr-form=1', *flags, '-I', os.path.join(ROOT, 'include'), *srcs,
               '-o', os.path.join(OUT, f'libmtr_rt_{name}.so')]
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
    for name, p in procs:
        _, err = p.communicate()
        if p.returncode:
            sys.exit(name + '\n' + err.decode())


def run_one(name):
    sys.path.insert(0, ROOT)
    import ctypes
    import torch
    from metrabs_amd import _lib
    lib = ctypes.CDLL(os.path.join(OUT, f'libmtr_rt_{name}.so'))
    hp = _lib.HeadParams(256, 32, 1, 0, 2200.0)
    g = torch.Generator(device='cuda').manual_seed(0)
    res = {'variant': name}
    lib.mtr_head_packed_bytes.restype = ctypes.c_size_t
    cases = [('B64', 64, 8, False, 8), ('B64 nhwc', 64, 8, True, 8), ('B1024', 1024, 8, False, 8),
             ('B32 12x12', 32, 12, False, 8), ('B256 12x12', 256, 12, False, 8),
             ('B64 D72', 64, 8, False, 72), ('B256 D72', 256, 8, False, 72), ('B256 8x8', 256, 8, False, 8),
             ('B16 24x24', 16, 24, False, 8), ('B64 16x16', 64, 16, False, 8)]
    if os.environ.get('RT_CASES'):
        cases = [c for c in cases if c[0] in os.environ['RT_CASES'].split(',')]
    for label, B, H, nhwc, D in cases:
        C, J = 1280, 17
        feat = torch.randn(B, C, H, H, device='cuda', generator=g)
        if nhwc:
            feat = feat.contiguous(memory_format=torch.channels_last)
        w = torch.randn(J * (1 + D), C, device='cuda', generator=g) * 0.03
        bias = torch.zeros(J * (1 + D), device='cuda')
        nb = lib.mtr_head_packed_bytes(C, J, D, 0)
        packed = torch.empty(nb // 4, device='cuda')
        vp = ctypes.c_void_p
        s = vp(torch.cuda.current_stream().cuda_stream)
        assert lib.mtr_head_pack_weights(vp(w.data_ptr()), vp(bias.data_ptr()), C, J, D, 0,
                                         vp(packed.data_ptr()), s) == 0
        c2 = torch.empty(B, J, 2, device='cuda'); c3 = torch.empty(B, J, 3, device='cuda')

        # RT_KGROUPS=1|2 in the environment: mtr_head_options.rt_k_groups for every call
        # RT_TILES / RT_LOADER / RT_SPLIT likewise
        opts = _lib.head_options(rt_tiles=int(os.environ.get('RT_TILES', '0')),
                                 rt_k_groups=int(os.environ.get('RT_KGROUPS', '0')),
                                 rt_loader=int(os.environ.get('RT_LOADER', '0')),
                                 rt_split=int(os.environ.get('RT_SPLIT', '0')))

        lib.mtr_head_workspace_bytes.restype = ctypes.c_size_t
        ws_bytes = lib.mtr_head_workspace_bytes(0, 1 if nhwc else 0, B, C, H, H, J, D) if os.environ.get('RT_WS') else 0
        ws = torch.empty(max(ws_bytes, 8) // 8 + 1, dtype=torch.float64, device='cuda')

        def call(stream):
            if ws_bytes:   # RT_WS=1: the default path of the Python wrappers (column blocks over workgroups)
                rc = lib.mtr_head_fused_ws(vp(feat.data_ptr()), 0, 1 if nhwc else 0, B, C, H, H,
                                           vp(packed.data_ptr()), J, D, ctypes.byref(hp), ctypes.byref(opts),
                                           vp(ws.data_ptr()), ctypes.c_size_t(ws.numel() * 8),
                                           vp(c2.data_ptr()), vp(c3.data_ptr()), vp(stream))
            else:
                rc = lib.mtr_head_fused_opts(vp(feat.data_ptr()), 0, 1 if nhwc else 0, B, C, H, H,
                                             vp(packed.data_ptr()), J, D, ctypes.byref(hp), ctypes.byref(opts),
                                             vp(c2.data_ptr()), vp(c3.data_ptr()), vp(stream))
            assert rc == 0, rc
        for _ in range(5):
            call(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        n = 20
        st = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            call(st.cuda_stream)
            st.synchronize()
            with torch.cuda.graph(graph, stream=st):
                for _ in range(n):
                    call(st.cuda_stream)
        graph.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        a.record()
        for _ in range(reps):
            graph.replay()
        b.record()
        torch.cuda.synchronize()
        res[label] = round(a.elapsed_time(b) / (n * reps) * 1e3, 1)
        if os.environ.get('RT_DUMP'):  # outputs beside the timings: `run` prints each variant's distance to 'full'
            os.makedirs(os.environ['RT_DUMP'], exist_ok=True)
            torch.save((c2.cpu(), c3.cpu()), os.path.join(os.environ['RT_DUMP'], f"{name}_{label.replace(' ', '_')}.pt"))
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    elif sys.argv[1] == 'run':
        for name in VARIANTS:
            subprocess.run([sys.executable, os.path.abspath(__file__), 'one', name], check=False)
        if os.environ.get('RT_DUMP'):
            import glob
            import torch
            for f in sorted(glob.glob(os.path.join(os.environ['RT_DUMP'], 'full_*.pt'))):
                ref = torch.load(f)
                for name in VARIANTS:
                    g = f.replace('full_', name + '_', 1)
                    if name != 'full' and os.path.exists(g):
                        out = torch.load(g)
                        print(os.path.basename(g), 'max |d coords2d| %.3e  max |d coords3d| %.3e' % (
                            (out[0] - ref[0]).abs().max(), (out[1] - ref[1]).abs().max()), flush=True)
    else:
        run_one(sys.argv[2])
