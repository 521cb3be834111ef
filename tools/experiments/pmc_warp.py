"""Launch the sampler a few times on the bench's shape (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd import _lib
if os.environ.get('PMC_WARP_LIB'):  # an ablation build of tools/experiments/ablate_warp.py
    _lib.load(os.environ['PMC_WARP_LIB'])
from metrabs_amd import kernels
from metrabs_amd.multiperson.multiperson_model import tta_parameters
num_aug = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator().manual_seed(0)
frames = torch.randint(0, 256, (8, 3, 1080, 1920), dtype=torch.uint8, generator=g).cuda()
pyr = kernels.build_pyramid(frames)
n = 64
tta = {k: v.cuda() for k, v in tta_parameters(num_aug).items()}
bw = 60 + 340 * torch.rand(n, generator=g)
bh = 150 + 750 * torch.rand(n, generator=g)
boxes = torch.stack([torch.rand(n, generator=g) * (1920 - bw),
                     torch.rand(n, generator=g) * (1080 - bh).clamp_min(1), bw, bh], 1).cuda()
K = torch.tensor([[1844.0, 0, 960], [0, 1844.0, 540], [0, 0, 1]]).repeat(n, 1, 1).cuda()
up = torch.tensor([0.0, -1, 0]).repeat(n, 1).cuda()
ids = (torch.arange(n) % 8).int().cuda()
_, _, wp = kernels.crop_geometry(boxes, K, torch.zeros(n, 12).cuda(), up, ids, tta['rotflipmat'],
                                 tta['scales'], tta['gammas'], 256, 1)
o = torch.empty(n * num_aug, 3, 256, 256, device='cuda')
for _ in range(5):
    kernels.warp_crops(pyr, wp, 256, 1, out=o)
torch.cuda.synchronize()
