import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrabs_amd import kernels
from metrabs_amd.config import MetrabsConfig
B = int(sys.argv[1]); dt = torch.float32 if sys.argv[2] == 'f32' else torch.float16
g = torch.Generator(device='cuda').manual_seed(0)
feat = torch.randn(B, 1280, 8, 8, device='cuda', generator=g).to(dt)
w = torch.randn(153, 1280, device='cuda', generator=g) * 0.03
packed = kernels.head_pack_weights(w, torch.zeros(153, device='cuda'), 17, 8, dt)
for _ in range(5):
    kernels.head_fused(feat, packed, 1280, 17, MetrabsConfig())
torch.cuda.synchronize()
