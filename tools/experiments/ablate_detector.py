"""Timing ablations of detector_pre_kernel (developer tool).  build here, run on the GPU box.
MTR_DET_ABLATE bits: 1 = no staging loads, 2 = no horizontal pass, 4 = no vertical pass, 8 = no final pow
(streaming kernel only).  DET_KERNEL=tile|stream|auto picks the kernel timed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tools', 'experiments', '_build')
MASKS = [int(m) for m in os.environ.get('DET_MASKS', '0,1,2,4,8,6,14,15').split(',')]


def build():
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, 'metrabs_amd', 'csrc')
    srcs = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith('.hip'))
    procs = [subprocess.Popen(['hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950',
                               f'-DMTR_DET_ABLATE={m}', '-I', os.path.join(ROOT, 'include'), *srcs, '-o',
                               os.path.join(OUT, f'libmtr_det{m}.so')], stdout=subprocess.DEVNULL,
                              stderr=subprocess.PIPE) for m in MASKS]
    for p in procs:
        _, err = p.communicate()
        if p.returncode:
            sys.exit(err.decode())


def run_one(mask):
    sys.path.insert(0, ROOT)
    import torch
    from metrabs_amd import _lib
    _lib.load(os.path.join(OUT, f'libmtr_det{mask}.so'))
    from metrabs_amd import kernels
    g = torch.Generator().manual_seed(0)
    res = {'mask': mask}
    for name, n, h, w in [('8x1080p', 8, 1080, 1920), ('8x480x640', 8, 480, 640)]:
        frames = torch.randint(0, 256, (n, 3, h, w), dtype=torch.uint8, generator=g).cuda()
        geom = kernels.detector_geometry(h, w)
        o = torch.empty(n, 3, geom.out_h, geom.out_w, device='cuda')
        for _ in range(5):
            kernels.detector_preprocess(frames, geom=geom, out=o, kernel=os.environ.get('DET_KERNEL', 'auto'))
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            kernels.detector_preprocess(frames, geom=geom, out=o, kernel=os.environ.get('DET_KERNEL', 'auto'))
        b.record()
        torch.cuda.synchronize()
        res[name] = round(a.elapsed_time(b) / 30 * 1e3, 1)
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    elif sys.argv[1] == 'run':
        for m in MASKS:
            subprocess.run([sys.executable, __file__, 'one', str(m)])
    else:
        run_one(int(sys.argv[2]))
