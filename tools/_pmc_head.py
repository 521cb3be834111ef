"""Launch the fused head a few times (for rocprofv3 --pmc passes): B dtype [J] [side] [nhwc]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrabs_amd import kernels
from metrabs_amd.config import MetrabsConfig
B = int(sys.argv[1]); dt = {'f32': torch.float32, 'f16': torch.float16, 'bf16': torch.bfloat16}[sys.argv[2]]
J = int(sys.argv[3]) if len(sys.argv) > 3 else 17
side = int(sys.argv[4]) if len(sys.argv) > 4 else 8
nhwc = len(sys.argv) > 5 and sys.argv[5] == 'nhwc'
g = torch.Generator(device='cuda').manual_seed(0)
feat = torch.randn(B, 1280, side, side, device='cuda', generator=g).to(dt)
if nhwc:
    feat = feat.contiguous(memory_format=torch.channels_last)
w = torch.randn(J * 9, 1280, device='cuda', generator=g) * 0.03
packed = kernels.head_pack_weights(w, torch.zeros(J * 9, device='cuda'), J, 8, dt)
cfg = MetrabsConfig(proc_side=side * 32)
opts = dict(dma_staging=int(os.environ['PMC_DMA'])) if os.environ.get('PMC_DMA') else {}   # (a forced 16-bit kernel variant)
for _ in range(5):
    kernels.head_fused(feat, packed, 1280, J, cfg, **opts)
torch.cuda.synchronize()
