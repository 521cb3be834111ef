"""Launch the detector pre-processing a few times on 8 x 1080p (for rocprofv3 --pmc passes).
    DET_KERNEL=stream|tile (default stream)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd import kernels
g = torch.Generator().manual_seed(0)
frames = torch.randint(0, 256, (8, 3, 1080, 1920), dtype=torch.uint8, generator=g).cuda()
geom = kernels.detector_geometry(1080, 1920)
o = torch.empty(8, 3, geom.out_h, geom.out_w, device='cuda')
for _ in range(6):
    kernels.detector_preprocess(frames, geom=geom, out=o, kernel=os.environ.get('DET_KERNEL', 'stream'))
torch.cuda.synchronize()
