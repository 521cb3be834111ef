"""Compile-time variants of the crop sampler (csrc/warp.hip), built HERE (hipcc cross-compiles) and timed on
the GPU box: only warp.hip is recompiled per variant and linked with the library's other objects.
    python tools/experiments/warp_variants.py build          # here
    python tools/experiments/warp_variants.py run            # on the GPU box: one JSON line per (variant, case)
Cases: 64 crops (num_aug 1) and 320 crops (64 boxes x 5 TTA), f32 crops of 8 x 1080p uint8 frames, the frames
rotating over 4 sets (> the Infinity Cache) -- the HBM-true number bench.py reports."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tools', 'experiments', '_build')
# (round 5, profiles/r05v_*: ablations of the shipped kernel; the unaligned-load and late-LUT variants measured there
#  were source edits that were not kept)
if os.environ.get('MTR_WARP_SET') == 'asm':   # round 6: counted waits (MTR_WARP_ASM) x samples requested ahead
    VARIANTS_R6 = {'builtin_waits_pd1': ['-DMTR_WARP_ASM=0'], 'asm_pd1': [], 'asm_pd2': ['-DMTR_WARP_PREFETCH=2'],
                   'asm_pd3': ['-DMTR_WARP_PREFETCH=3'], 'builtin_waits_pd2': ['-DMTR_WARP_ASM=0', '-DMTR_WARP_PREFETCH=2']}
VARIANTS = {'base': [], 'one_gather_pair': ['-DMTR_WARP_ABLATE=16'], 'no_taps': ['-DMTR_WARP_ABLATE=1'],
            'no_stores': ['-DMTR_WARP_ABLATE=8'], 'arithmetic_only': ['-DMTR_WARP_ABLATE=15']}


if os.environ.get('MTR_WARP_SET') == 'asm':
    VARIANTS = VARIANTS_R6
if os.environ.get('MTR_WARP_SET') == 'pyr':   # round 6: the pyramid's LUT copies (LDS per workgroup -> workgroups per CU)
    VARIANTS = {'lut32': ['-DMTR_PYR_LUT_COPIES=32'], 'lut16': ['-DMTR_PYR_LUT_COPIES=16'], 'lut8': ['-DMTR_PYR_LUT_COPIES=8'],
                'lut16_all_wgs': ['-DMTR_PYR_LUT_COPIES=16', '-DMTR_PYR_PER_CU=16'], 'lut8_all_wgs': ['-DMTR_PYR_LUT_COPIES=8', '-DMTR_PYR_PER_CU=16'],
                'lut16_6': ['-DMTR_PYR_LUT_COPIES=16', '-DMTR_PYR_PER_CU=6']}
if os.environ.get('MTR_WARP_SET') == 'waves':   # round 6: waves per workgroup
    VARIANTS = {'waves4': [], 'waves1': ['-DMTR_WARP_WAVES=1'], 'waves2': ['-DMTR_WARP_WAVES=2'], 'waves8': ['-DMTR_WARP_WAVES=8'],
                'waves1_rows8': ['-DMTR_WARP_WAVES=1', '-DMTR_WARP_ROWS=8'], 'waves2_rows8': ['-DMTR_WARP_WAVES=2', '-DMTR_WARP_ROWS=8']}
if os.environ.get('MTR_WARP_SET') == 'rows':   # round 6: fewer samples per wave (more, shorter waves)
    VARIANTS = {'rows4': [], 'rows2': ['-DMTR_WARP_ROWS=2'], 'rows1': ['-DMTR_WARP_ROWS=1'], 'rows3': ['-DMTR_WARP_ROWS=3'],
                'rows2_lx64': ['-DMTR_WARP_ROWS=2', '-DMTR_WARP_LX=64']}


def build():
    sys.path.insert(0, ROOT)
    from metrabs_amd import build as b
    b.build_library(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    others = [os.path.join(b.BUILD_DIR, f + '.o') for f in b.sources() if f != 'warp.hip']
    procs = []
    for name, defs in VARIANTS.items():
        obj = os.path.join(OUT, f'warp_{name}.o')
        procs.append((name, obj, subprocess.Popen([b._hipcc(), *b.FLAGS, *defs, '-c', os.path.join(b.CSRC, 'warp.hip'),
                                                   '-o', obj], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
        if len(procs) % 6 == 0:
            for _, _, p in procs[-6:]:
                p.wait()
    for name, obj, p in procs:
        _, err = p.communicate()
        if p.returncode:
            sys.exit(f'{name}: {err.decode()[-2000:]}')
        subprocess.run([b._hipcc(), '-shared', '-fPIC', f'--offload-arch={b.ARCH}', *others, obj, '-o',
                        os.path.join(OUT, f'libmtr_warp_{name}.so')], check=True)
    print('built', len(VARIANTS), 'variants')


def run_one(name, interleaved=False):
    sys.path.insert(0, ROOT)
    import torch
    from metrabs_amd import _lib
    _lib.load(os.path.join(OUT, f'libmtr_warp_{name}.so'))
    from metrabs_amd import kernels
    from metrabs_amd.multiperson.multiperson_model import tta_parameters
    g = torch.Generator().manual_seed(0)
    fmt = torch.channels_last if interleaved else torch.contiguous_format
    pyrs = [kernels.build_pyramid(torch.randint(0, 256, (8, 3, 1080, 1920), dtype=torch.uint8, generator=g).cuda()
                                  .contiguous(memory_format=fmt)) for _ in range(4)]
    assert all(p.hwc == interleaved for p in pyrs)
    if os.environ.get('MTR_WARP_SET') == 'pyr':   # the pyramid pass itself, frames in rotation (six sets: 300 MB of uint8)
        import hashlib
        from tools.microbench import timeit
        frames = [torch.randint(0, 256, (8, 3, 1080, 1920), dtype=torch.uint8, generator=g).cuda().contiguous(memory_format=fmt)
                  for _ in range(6)]
        k = [0]

        def step():
            k[0] += 1
            return kernels.build_pyramid(frames[k[0] % 6])
        t = min(timeit(step, iters=30) for _ in range(3))
        pyr = kernels.build_pyramid(frames[0])
        lv = [x for x in (getattr(pyr, 'levels', None) or []) if x is not None and x.dtype == torch.float32]
        h = hashlib.sha1(b''.join(x.cpu().numpy().tobytes() for x in lv[-2:])).hexdigest()[:12] if lv else ''
        print(json.dumps({'variant': name, 'frames': 'interleaved' if interleaved else 'planar', 'kernel': 'build_pyramid', 'us': round(t * 1e6, 2), 'sha': h}), flush=True)
        return
    for aug in (1, 5):
        n = 64
        tta = {k: v.cuda() for k, v in tta_parameters(aug).items()}
        gb = torch.Generator().manual_seed(1)
        bw = 60 + 340 * torch.rand(n, generator=gb)
        bh = 150 + 750 * torch.rand(n, generator=gb)
        boxes = torch.stack([torch.rand(n, generator=gb) * (1920 - bw),
                             torch.rand(n, generator=gb) * (1080 - bh).clamp_min(1), bw, bh], 1).cuda()
        K = torch.tensor([[1844.0, 0, 960], [0, 1844.0, 540], [0, 0, 1]]).repeat(n, 1, 1).cuda()
        up = torch.tensor([0.0, -1, 0]).repeat(n, 1).cuda()
        ids = (torch.arange(n) % 8).int().cuda()
        _, _, wp = kernels.crop_geometry(boxes, K, torch.zeros(n, 12).cuda(), up, ids, tta['rotflipmat'],
                                         tta['scales'], tta['gammas'], 256, 1)
        o = torch.empty(n * aug, 3, 256, 256, device='cuda')
        for _ in range(3):
            kernels.warp_crops(pyrs[0], wp, 256, 1, out=o)
        torch.cuda.synchronize()
        reps = 20
        st = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            kernels.warp_crops(pyrs[0], wp, 256, 1, out=o)
            st.synchronize()
            with torch.cuda.graph(graph, stream=st):
                for i in range(reps):
                    kernels.warp_crops(pyrs[i % len(pyrs)], wp, 256, 1, out=o)
        graph.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            graph.replay()
        b.record()
        torch.cuda.synchronize()
        print(json.dumps({'variant': name, 'frames': 'interleaved' if interleaved else 'planar', 'crops': n * aug, 'us': round(a.elapsed_time(b) / (reps * 10) * 1e3, 2),
                          'checksum': float(o.double().sum())}), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    elif sys.argv[1] == 'run':
        for m in list(VARIANTS) * (2 if os.environ.get('MTR_WARP_SET') else 1):
            for layout in ('planar', 'interleaved'):
                subprocess.run([sys.executable, __file__, 'one', m, layout])
    else:
        run_one(sys.argv[2], len(sys.argv) > 3 and sys.argv[3] == 'interleaved')
