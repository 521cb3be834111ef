"""The hand-written kernels of one bench step at the bench's shapes, launched a few times each with
nothing else around them -- the command the two rocprofv3 --pmc passes of bench.py:live_pmc_traffic
run (FETCH_SIZE, WRITE_SIZE; --kernel-trace only).  Sampler kernels work on a new set of frames per
launch (sets spanning more than the 256 MiB Infinity Cache), the head on rotating feature maps, the
stand-alone decode at its 1.28 GB roofline shape.

    python tools/_pmc_step.py n_crops res precision J D C frames num_aug
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402

n_crops, res = int(sys.argv[1]), int(sys.argv[2])
dt = {'f32': torch.float32, 'f16': torch.float16, 'bf16': torch.bfloat16}[sys.argv[3]]
J, D, C, frames, num_aug = (int(a) for a in sys.argv[4:9])
im_h, im_w = 1080, 1920
dev = torch.device('cuda')
g = torch.Generator(device='cuda').manual_seed(11)
cfg = MetrabsConfig(proc_side=res, depth=D)
n_box = n_crops // num_aug

# sampler inputs as bench.py:synth_inputs builds them
gc = torch.Generator().manual_seed(100)
bw = 60 + 340 * torch.rand(n_box, generator=gc)
bh = 150 + 750 * torch.rand(n_box, generator=gc)
bx = torch.rand(n_box, generator=gc) * (im_w - bw)
by = torch.rand(n_box, generator=gc) * (im_h - bh).clamp_min(1.0)
boxes = torch.stack([bx, by, bw, bh], dim=1).to(dev)
f = max(im_h, im_w) / (np.tan(np.deg2rad(55.0) / 2) * 2)
K = torch.tensor([[f, 0, im_w / 2], [0, f, im_h / 2], [0, 0, 1]], dtype=torch.float32).repeat(n_box, 1, 1).to(dev)
ids = (torch.arange(n_box) % frames).int().to(dev)
dist12 = torch.zeros(n_box, 12, device=dev)
up = torch.tensor([0.0, -1.0, 0.0], device=dev).repeat(n_box, 1)

from metrabs_amd.multiperson.multiperson_model import tta_parameters  # noqa: E402
tta = {k: v.to(dev) for k, v in tta_parameters(num_aug).items()}

frame_bytes = frames * 3 * im_h * im_w
n_sets = max(2, -(-(640 << 20) // (frame_bytes + frame_bytes // 3)))
with torch.inference_mode():
    new_k, rot, wp = kernels.crop_geometry(boxes, K, dist12, up, ids, tta['rotflipmat'], tta['scales'],
                                           tta['gammas'], res, 1)
    sets = [torch.randint(0, 256, (frames, 3, im_h, im_w), dtype=torch.uint8, device=dev, generator=g)
            for _ in range(n_sets)]
    for rep in range(2):
        for fr in sets:
            pyr = kernels.build_pyramid(fr)
            kernels.warp_crops(pyr, wp, res, 1, out_dtype=dt)
    del sets
    hw = res // 32
    n_feat = max(2, min(16, -(-(640 << 20) // (n_crops * C * hw * hw * (4 if dt == torch.float32 else 2)))))
    feats = [torch.randn(n_crops, C, hw, hw, device=dev, generator=g).to(dt) for _ in range(n_feat)]
    w = torch.randn(J * (1 + D), C, device=dev, generator=g) * 0.03
    packed = kernels.head_pack_weights(w, torch.zeros(J * (1 + D), device=dev), J, D, dt)
    for rep in range(2):
        for ft in feats:
            c2d, c3d = kernels.head_fused(ft, packed, C, J, cfg)
    kflat = new_k.reshape(-1, 3, 3)
    for rep in range(4):
        poses = kernels.reconstruct_absolute(c2d, c3d, kflat, cfg)
    del feats
    if os.environ.get('MTR_PMC_SKIP_DECODE') != '1':
        logits = torch.randn(32768, 17 * 9, 8, 8, device=dev, generator=g)
        for rep in range(3):
            kernels.softargmax_decode(logits, 17, MetrabsConfig())
torch.cuda.synchronize()
