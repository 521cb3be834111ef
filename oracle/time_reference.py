"""TEST INFRASTRUCTURE: the REAL reference (metrabs_pytorch, run in place through oracle/ref_harness.py)
timed beside the in-repo restatement (oracle/cpu_ref.py) on the SAME synthetic workload, in the build
container (where /root/reference is mounted; the GPU box has no reference).  bench.py's
`cpu_baseline.kind` is "port" (the restatement); this script is the evidence that the port and the
reference cost the same on the CPU.

    python -m oracle.time_reference [n_frames=1] [boxes_per_frame=8] [threads=...]

Workload = bench.py's config 1 per frame: 1080p uint8 frames, 8 boxes each, EfficientNetV2-S
(the reference's own class with random weights, batch-norm statistics as initialised), 256 px crops,
num_aug 1, through Pose3dEstimator._estimate_poses_batched on both sides.
"""
import sys
import time

import numpy as np
import torch

from oracle import cases, cpu_ref, ref_harness as rh


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    per_frame = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    threads = [int(t) for t in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, torch.get_num_threads()]
    ref = rh.load()
    torch.manual_seed(0)
    with rh.config():
        net = ref.efficientnet.efficientnet_v2_s()
    backbone = torch.nn.Sequential(ref.efficientnet.PreprocLayer(), net.features).eval()
    ji = rh._JointInfoStub(cases.COCO17, cases.COCO17_EDGES)
    g = cases.gen(5)
    images = torch.randint(0, 256, (n_frames, 3, 1080, 1920), dtype=torch.uint8, generator=g)
    boxes = []
    for _ in range(n_frames):
        bw = 60 + 340 * torch.rand(per_frame, generator=g)
        bh = 150 + 750 * torch.rand(per_frame, generator=g)
        bx = torch.rand(per_frame, generator=g) * (1920 - bw)
        by = torch.rand(per_frame, generator=g) * (1080 - bh).clamp_min(1.0)
        boxes.append(torch.stack([bx, by, bw, bh, torch.ones(per_frame)], dim=1))
    w, b = cases.default_conv_init(17 * 9, 1280, g)
    with rh.config(), torch.inference_mode():
        crop_model = ref.metrabs_model.Metrabs(backbone, ji).eval()
        conv = torch.nn.Conv2d(1280, 153, 1)
        conv.weight.copy_(w[:, :, None, None])
        conv.bias.copy_(b)
        crop_model.heatmap_heads.conv_final = conv
        skel = {'': dict(indices=list(range(17)), names=cases.COCO17, edges=[[0, 1]])}
        est = ref.multiperson_model.Pose3dEstimator(crop_model, skel, np.eye(17, dtype=np.float32))
        est.joint_transform_matrix = None
    K = torch.full((1, 3, 3), -1.0)
    args = (torch.zeros(1, 5), torch.eye(4)[None], torch.tensor([0.0, -1.0, 0.0]))
    ocfg = cpu_ref.HeadConfig()
    mirror = cases.mirror_mapping(cases.COCO17)

    def port_crop_model(inp):
        crops, k = inp
        return cpu_ref.crop_model_from_features(backbone(crops), w, b, k, 17, ocfg)

    def run_reference():
        with rh.config(), torch.inference_mode():
            return est._estimate_poses_batched(images, boxes, K, *args, 55, 64, 1, 1, True, '', False)

    def run_port():
        with torch.inference_mode():
            return cpu_ref.estimate_poses_batched(port_crop_model, mirror, 17, 256, images, boxes, K, *args,
                                                  55, 64, 1, 1, True)

    n_crops = n_frames * per_frame
    for t in threads:
        torch.set_num_threads(t)
        res = {}
        for name, fn in (('reference', run_reference), ('port', run_port)):
            fn()
            t0 = time.time()
            out = fn()
            res[name] = (n_crops / (time.time() - t0), torch.cat(out['poses3d']))
        d = float((res['reference'][1] - res['port'][1]).abs().max())
        print(f'threads {t}: reference {res["reference"][0]:.2f} crops/s, port {res["port"][0]:.2f} crops/s '
              f'({n_crops} crops of {n_frames} 1080p frame(s)); max |reference - port| = {d:.2e} mm', flush=True)


if __name__ == '__main__':
    main()
