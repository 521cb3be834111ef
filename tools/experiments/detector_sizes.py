import sys, json, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
from metrabs_amd import kernels
from microbench import timeit
g = torch.Generator().manual_seed(0)
for n, h, w in [(1,1080,1920),(2,1080,1920),(3,1080,1920),(4,1080,1920),(8,1080,1920),(16,1080,1920),(1,480,640),(2,480,640),(4,480,640),(8,480,640),(1,720,1280),(4,720,1280),(1,2160,3840),(1,256,416),(8,256,416)]:
    frames = torch.randint(0, 256, (n, 3, h, w), dtype=torch.uint8, generator=g).cuda()
    geom = kernels.detector_geometry(h, w)
    o = torch.empty(n, 3, geom.out_h, geom.out_w, device='cuda')
    r = {}
    for kern in ('stream', 'tile'):
        r[kern] = round(timeit(lambda: kernels.detector_preprocess(frames, geom=geom, out=o, kernel=kern)) * 1e6, 1)
    print(json.dumps(dict(n=n, h=h, w=w, MB=round(frames.numel()/1e6,1), **r)), flush=True)
