"""Degenerate inputs through the public API: nothing may crash or hang (values may be NaN where the
reference's would be).  Developer probe."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import cases
from test_gpu_e2e import build_estimator
case = cases.e2e_case('aug5')
est = build_estimator(case, True)
img = case['images']
h, w = img.shape[2:]
weird = [
    ('zero-size box', [torch.tensor([[10.0, 10.0, 0.0, 0.0, 1.0]])] + [torch.zeros(0, 5)] * (len(img) - 1)),
    ('box outside the frame', [torch.tensor([[5000.0, -3000.0, 50.0, 80.0, 1.0]])] + [torch.zeros(0, 5)] * (len(img) - 1)),
    ('huge box', [torch.tensor([[-1e4, -1e4, 3e4, 3e4, 1.0]])] + [torch.zeros(0, 5)] * (len(img) - 1)),
    ('negative size', [torch.tensor([[50.0, 40.0, -30.0, -60.0, 1.0]])] + [torch.zeros(0, 5)] * (len(img) - 1)),
    ('nan box', [torch.tensor([[float('nan'), 1.0, 20.0, 30.0, 1.0]])] + [torch.zeros(0, 5)] * (len(img) - 1)),
    ('no boxes at all', [torch.zeros(0, 5)] * len(img)),
    ('many boxes', [torch.cat([case['boxes'][0]] * 40)] + [torch.zeros(0, 5)] * (len(img) - 1)),
]
for name, boxes in weird:
    for num_aug in (1, 5):
        with torch.inference_mode():
            r = est._estimate_poses_batched(img, boxes, case['K'], case['dist'], case['extr'], case['world_up'],
                                            55, 7, 1, num_aug, True, '', False)
        torch.cuda.synchronize()
        p = torch.cat(r['poses3d'])
        print(f'{name:24s} aug={num_aug}: {tuple(p.shape)} finite={bool(torch.isfinite(p).all()) if p.numel() else "-"}')
# singular / NaN intrinsics
K = case['K'].clone(); K[0] = 0
with torch.inference_mode():
    r = est._estimate_poses_batched(img, case['boxes'], K, case['dist'], case['extr'], case['world_up'], 55, 64, 1, 2, True, '', False)
torch.cuda.synchronize(); print('singular K ok', tuple(torch.cat(r['poses3d']).shape))
print('done')

# the same degenerate boxes through the oracle: NaN for NaN, numbers for numbers
from oracle import cpu_ref
mm = cases.mirror_mapping(cases.COCO17)
backbone_cpu = cases.e2e_case('aug5')['backbone']
def crop_model(inp):
    crops, K = inp
    return cpu_ref.crop_model_from_features(backbone_cpu(crops), case['head_w'], case['head_b'], K, 17, case['cfg'])
for name, boxes in weird[:5]:
    for num_aug in (1, 5):
        with torch.inference_mode():
            ours = est._estimate_poses_batched(img, boxes, case['K'], case['dist'], case['extr'], case['world_up'],
                                               55, 7, 1, num_aug, True, '', False)
            try:
                ref = cpu_ref.estimate_poses_batched(crop_model, mm, 17, case['res'], img, boxes, case['K'],
                                                     case['dist'], case['extr'], case['world_up'], 55, 7, 1, num_aug, True)
            except Exception as e:  # the reference fails on these (lstsq on NaN, level index from NaN)
                print(f'{name:24s} aug={num_aug}: oracle raises ({str(e)[:60]}...)')
                continue
        a, b = torch.cat(ours['poses3d']).cpu(), torch.cat(ref['poses3d'])
        fa, fb = torch.isfinite(a), torch.isfinite(b)
        both = fa & fb
        d = float((a[both] - b[both]).abs().max()) if both.any() else float('nan')
        print(f'{name:24s} aug={num_aug}: finite ours {float(fa.float().mean()):.2f} oracle {float(fb.float().mean()):.2f} max diff on common {d:.3e}')
