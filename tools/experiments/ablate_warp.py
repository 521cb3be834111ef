"""Timing ablations of warp_crops_kernel (developer tool).  build here, run on the GPU box.
MTR_WARP_ABLATE bits: 1 = no tap loads (uint8 path), 2 = no LUT lookups, 4 = no gamma pow,
8 = no output stores, 16 = one gather pair serves all three channels of a level-0 pixel.
Round 2 (64 crops): full 34.1 us; 16: 29.6; 1: 29.2; 8: 24.6; 15: 22.1 -- a third of the gathers buys
4.5 us, i.e. a channel-interleaved pyramid (2 gathers per pixel instead of 6) would cost more in the
pyramid kernel than it saves here; non-temporal crop stores: 31.5 -> 30.1 us (not kept: the backbone
reads the crops next)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tools', 'experiments', '_build')
MASKS = [0, 16, 1, 2, 4, 8, 15]  # 16 = one gather pair serves all three channels (level-0 crops)
# variant name -> extra defines (MASKS entries are the variants 'a<mask>')
VARIANTS = {f'{m}': [f'-DMTR_WARP_ABLATE={m}', '-DMTR_WARP_ROWS=0'] for m in MASKS}
VARIANTS.update({f'rows{r}': [f'-DMTR_WARP_ROWS={r}'] for r in (1, 2, 4, 8, 16)})
VARIANTS.update({'pd1': [], 'pd2': ['-DMTR_WARP_PREFETCH=2'], 'pd3': ['-DMTR_WARP_PREFETCH=3'], 'pd2r8': ['-DMTR_WARP_PREFETCH=2', '-DMTR_WARP_ROWS=8'], 'pd0': ['-DMTR_WARP_PREFETCH=0']})
VARIANTS.update({'lx64': [], 'lx32': ['-DMTR_WARP_LX=32'], 'lx16': ['-DMTR_WARP_LX=16'], 'lx32r2': ['-DMTR_WARP_LX=32', '-DMTR_WARP_ROWS=2'], 'lx16r2': ['-DMTR_WARP_LX=16', '-DMTR_WARP_ROWS=2'], 'lx16r1': ['-DMTR_WARP_LX=16', '-DMTR_WARP_ROWS=1']})
VARIANTS.update({'rcp': ['-DMTR_WARP_RCP=1'], 'rcp8': ['-DMTR_WARP_RCP=1', '-DMTR_WARP_ROWS=8']})
VARIANTS.update({'px8': ['-DMTR_WARP_PX=8'], 'px8_nostore': ['-DMTR_WARP_PX=8', '-DMTR_WARP_ABLATE=8'],
                 'px16': ['-DMTR_WARP_PX=16'], 'px2': ['-DMTR_WARP_PX=2'], 'px1': ['-DMTR_WARP_PX=1'],
                 'px2_nomem': ['-DMTR_WARP_PX=2', '-DMTR_WARP_ABLATE=15'], 'px8_nomem': ['-DMTR_WARP_PX=8', '-DMTR_WARP_ABLATE=15']})
VARIANTS.update({f'r{m}': [f'-DMTR_WARP_ABLATE={m}'] for m in (0, 1, 2, 4, 8, 3, 15)})  # the same bits in warp_rows_kernel
VARIANTS.update({'r0_fat': ['-DMTR_WARP_LEAN=0']})
# round 6: the persistent launch (workgroups per CU), alone and with longer bands / deeper prefetch per wave
VARIANTS.update({f'persist{n}': [f'-DMTR_WARP_PERSIST={n}'] for n in (1, 2, 4, 8)})
VARIANTS.update({'persist4r8': ['-DMTR_WARP_PERSIST=4', '-DMTR_WARP_ROWS=8'], 'persist8pd2': ['-DMTR_WARP_PERSIST=8', '-DMTR_WARP_PREFETCH=2'],
                 'persist4pd3': ['-DMTR_WARP_PERSIST=4', '-DMTR_WARP_PREFETCH=3']})
if os.environ.get('ABLATE_ONLY'):
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in os.environ['ABLATE_ONLY'].split(',')}


def build():
    """Every variant = warp.hip recompiled with its defines + the product build's objects of the other sources
    (metrabs_amd/csrc/build/*.o, made by `python -m metrabs_amd.build`): seconds per variant instead of a whole-
    library compile each (round 6)."""
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    from metrabs_amd import build as product
    product.build_library(verbose=False)
    others = [os.path.join(product.BUILD_DIR, f + '.o') for f in product.sources() if f != 'warp.hip']
    src = os.path.join(product.CSRC, 'warp.hip')
    items = list(VARIANTS.items())
    for i in range(0, len(items), 8):
        procs = []
        for m, defs in items[i:i + 8]:
            obj = os.path.join(OUT, f'warp_{m}.o')
            procs.append((m, obj, subprocess.Popen(['hipcc', *product.FLAGS, *defs, '-c', src, '-o', obj],
                                                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
        for m, obj, p in procs:
            _, err = p.communicate()
            if p.returncode:
                sys.exit(err.decode()[-3000:])
            r = subprocess.run(['hipcc', '-shared', '-fPIC', f'--offload-arch={product.ARCH}', obj, *others, '-o',
                                os.path.join(OUT, f'libmtr_warp{m}.so')], capture_output=True, text=True)
            if r.returncode:
                sys.exit(r.stderr[-3000:])
            os.remove(obj)


def run_one(mask):
    sys.path.insert(0, ROOT)
    import torch
    from metrabs_amd import _lib
    _lib.load(os.path.join(OUT, f'libmtr_warp{mask}.so'))
    from metrabs_amd import kernels
    from metrabs_amd.multiperson.multiperson_model import tta_parameters
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (8, 3, 1080, 1920), dtype=torch.uint8, generator=g).cuda()
    pyr = kernels.build_pyramid(frames)
    # ABLATE_ROTATE=k: k frame sets (112 MB each with their pyramids), launches cycle through them
    # so that the gathers miss the Infinity Cache
    pyrs = [pyr] + [kernels.build_pyramid(torch.randint(0, 256, (8, 3, 1080, 1920), dtype=torch.uint8, generator=g).cuda())
                    for _ in range(int(os.environ.get('ABLATE_ROTATE', '1')) - 1)]
    n = int(os.environ.get('ABLATE_CROPS', '64'))
    aug = int(os.environ.get('ABLATE_AUG', '1'))  # 5 = the TTA table (rotations of +-25 degrees, flips, scales)
    n //= aug
    tta = {k: v.cuda() for k, v in tta_parameters(aug).items()}
    bw = 60 + 340 * torch.rand(n, generator=g)
    bh = 150 + 750 * torch.rand(n, generator=g)
    boxes = torch.stack([torch.rand(n, generator=g) * (1920 - bw),
                         torch.rand(n, generator=g) * (1080 - bh).clamp_min(1), bw, bh], 1).cuda()
    K = torch.tensor([[1844.0, 0, 960], [0, 1844.0, 540], [0, 0, 1]]).repeat(n, 1, 1).cuda()
    up = torch.tensor([0.0, -1, 0]).repeat(n, 1).cuda()
    ids = (torch.arange(n) % 8).int().cuda()
    _, _, wp = kernels.crop_geometry(boxes, K, torch.zeros(n, 12).cuda(), up, ids, tta['rotflipmat'],
                                     tta['scales'], tta['gammas'], 256, 1)
    o = torch.empty(n * aug, 3, 256, 256, device='cuda')
    for _ in range(5):
        kernels.warp_crops(pyr, wp, 256, 1, out=o)
    torch.cuda.synchronize()
    # 20 launches per graph replay: GPU time per launch without the Python / ctypes call floor
    n = 20
    st = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        kernels.warp_crops(pyr, wp, 256, 1, out=o)
        st.synchronize()
        with torch.cuda.graph(graph, stream=st):
            for i in range(n):
                kernels.warp_crops(pyrs[i % len(pyrs)], wp, 256, 1, out=o)
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    import hashlib
    print(json.dumps({'variant': mask, 'us': round(a.elapsed_time(b) / (n * 10) * 1e3, 1),
                      'checksum': float(o.double().sum()),
                      'sha256_16': hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16]}), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    elif sys.argv[1] == 'run':
        for m in VARIANTS:
            subprocess.run([sys.executable, __file__, 'one', str(m)])
    else:
        run_one(sys.argv[2])
