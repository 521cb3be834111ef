"""GPU: seeded shape fuzzing of the arithmetic kernels against the oracle (same bounds as the
fixed-case tests).  Shapes are drawn from a fixed seed, so failures are reproducible."""
import pytest
import torch

from oracle import cases, cpu_ref

pytestmark = pytest.mark.gpu


def mcfg(cfg):
    from metrabs_amd.config import MetrabsConfig
    return MetrabsConfig.from_any(cfg.as_dict())


def _rand_shapes(n, seed):
    g = cases.gen(seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    return [dict(B=r(1, 9), J=r(1, 40), D=r(1, 20), H=r(1, 20), W=r(1, 20), C=r(1, 12) * 8 + r(0, 7),
                 sigma=[0.5, 2.0, 8.0][r(0, 2)], i=i) for i in range(n)]


@pytest.mark.parametrize('sh', _rand_shapes(24, 1234), ids=lambda s: f"{s['i']}")
def test_decode_fuzz(sh, hip_lib):
    from metrabs_amd import kernels
    B, J, D, H, W = sh['B'], sh['J'], sh['D'], sh['H'], sh['W']
    cfg = cpu_ref.HeadConfig(depth=D, proc_side=max(H, W) * 8, stride_test=8, stride_train=8)
    g = cases.gen(5000 + sh['i'])
    logits = torch.randn(B, J * (1 + D), H, W, generator=g) * sh['sigma']
    with torch.inference_mode():
        o2d, o3d = cpu_ref.heads_from_logits(logits, J, cfg)
    c2d, c3d = kernels.softargmax_decode(logits.cuda(), J, mcfg(cfg))
    assert float((c3d.cpu() - o3d).abs().max()) <= 1e-3, sh
    assert float((c2d.cpu() - o2d).abs().max()) <= 2e-4, sh


@pytest.mark.parametrize('sh', [s for s in _rand_shapes(40, 4321)
                                if (s['H'] * s['W']) % 4 == 0 and s['H'] * s['W'] <= 256 and s['D'] < 63][:14],
                         ids=lambda s: f"{s['i']}")
def test_fused_head_fuzz(sh, hip_lib):
    from metrabs_amd import kernels
    B, J, D, H, W, C = sh['B'], sh['J'], sh['D'], sh['H'], sh['W'], sh['C']
    cfg = cpu_ref.HeadConfig(depth=D, proc_side=max(H, W) * 8, stride_test=8, stride_train=8)
    g = cases.gen(6000 + sh['i'])
    feat = torch.randn(B, C, H, W, generator=g)
    w, b = cases.default_conv_init(J * (1 + D), C, g)
    w, b = w * 3, b * 3
    with torch.inference_mode():
        o2d, o3d = cpu_ref.heads_forward(feat, w, b, J, cfg)
    packed = kernels.head_pack_weights(w.cuda(), b.cuda(), J, D)
    c2d, c3d = kernels.head_fused(feat.cuda(), packed, C, J, mcfg(cfg))
    assert float((c3d.cpu() - o3d).abs().max()) <= 2e-3, sh
    assert cpu_ref.mpjpe(c3d.cpu(), o3d) <= 1e-3, sh
    assert float((c2d.cpu() - o2d).abs().max()) <= 4e-4, sh


@pytest.mark.parametrize('i', range(10))
def test_reconstruct_fuzz(i, hip_lib):
    from metrabs_amd import kernels
    g = cases.gen(7000 + i)
    B = int(torch.randint(1, 90, (1,), generator=g))
    J = int(torch.randint(3, 140, (1,), generator=g))
    P = [256, 384, 160][i % 3]
    cfg = cpu_ref.HeadConfig(proc_side=P, centered_stride=bool(i % 2),
                             mix_3d_inside_fov=[0.5, 0.3, None][i % 3])
    f = (450 + 100 * torch.rand(B, generator=g)) * P / 256
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = f, f * 1.01, P / 2, P / 2 + 1, 1
    K[:, 0, 1] = 0.3
    ref = torch.stack([150 * torch.randn(B, generator=g), 150 * torch.randn(B, generator=g),
                       2000 + 3000 * torch.rand(B, generator=g)], dim=1)
    rel = torch.randn(B, J, 3, generator=g) * torch.tensor([300.0, 400.0, 250.0])
    abs3d = rel + ref[:, None]
    c2d = (abs3d[..., :2] / abs3d[..., 2:]) * f[:, None, None] + P / 2 + 2 * torch.randn(B, J, 2, generator=g)
    with torch.inference_mode():
        o = cpu_ref.reconstruct_absolute(c2d, rel, K, cfg)
    ours = kernels.reconstruct_absolute(c2d.cuda(), rel.cuda(), K.cuda(), mcfg(cfg)).cpu()
    assert cpu_ref.mpjpe(ours, o) <= 1e-3 and float((ours - o).abs().max()) <= 5e-3, (B, J, P)


def _e2e_fuzz_cfgs(n, seed):
    import random
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        out.append(dict(
            seed=9100 + i, n_images=rnd.randint(1, 3), imh=rnd.choice([96, 121, 150, 240]),
            imw=rnd.choice([128, 161, 200, 320]), res=rnd.choice([32, 64]), num_aug=rnd.randint(1, 6),
            aa=rnd.choice([1, 1, 2, 4]), dist=rnd.choice([None, cases.DISTORTION_5, cases.DISTORTION_12]),
            ibs=rnd.choice([2, 5, 64]), average_aug=rnd.random() < 0.5, extr=rnd.random() < 0.5,
            skeleton=rnd.random() < 0.5, jtm=rnd.random() < 0.5, known_k=rnd.random() < 0.7, i=i))
    return out


@pytest.mark.parametrize('c', _e2e_fuzz_cfgs(10, 77), ids=lambda c: f"{c['i']}")
def test_estimator_fuzz_vs_oracle(c, hip_lib):
    """Whole path from images (tiny backbone): random frame sizes (odd widths: the scalar pyramid and
    byte-tail paths), 1-6 augmentations, antialias 1/2/4, 0/5/12 distortion coefficients, internal
    batches that split a box's augmentations, extrinsics, skeleton selection, joint transform,
    unknown intrinsics -- against the oracle's restatement of _estimate_poses_batched on the CPU."""
    from test_gpu_e2e import build_estimator
    name = f"fuzz{c['i']}"
    cases.E2E_CASES[name] = {k: v for k, v in c.items() if k != 'i'}
    try:
        case = cases.e2e_case(name)
        backbone_cpu = cases.e2e_case(name)['backbone']
    finally:
        del cases.E2E_CASES[name]
    est = build_estimator(case, fused_head=True)
    with torch.inference_mode():
        ours = est._estimate_poses_batched(
            case['images'], case['boxes'], case['K'], case['dist'], case['extr'], case['world_up'],
            55, case['ibs'], case['aa'], case['num_aug'], case['average_aug'], '', False)
        mm = cases.mirror_mapping(cases.COCO17)

        def crop_model(inp):
            crops, K = inp
            return cpu_ref.crop_model_from_features(
                backbone_cpu(crops), case['head_w'], case['head_b'], K, 17, case['cfg'])

        ref = cpu_ref.estimate_poses_batched(
            crop_model, mm, 17, case['res'], case['images'], case['boxes'], case['K'], case['dist'],
            case['extr'], case['world_up'], 55, case['ibs'], case['aa'], case['num_aug'],
            case['average_aug'], joint_transform_matrix=case['jtm'], skeleton_indices=case['skeleton'])
    assert [len(p) for p in ours['poses3d']] == [len(p) for p in ref['poses3d']]
    p3, r3 = torch.cat(ours['poses3d']).cpu(), torch.cat(ref['poses3d'])
    assert p3.shape == r3.shape
    if p3.numel():
        print(f"[parity] e2e fuzz {c['i']}: MPJPE {cpu_ref.mpjpe(p3, r3):.2e} mm max {float((p3 - r3).abs().max()):.2e}")
        assert cpu_ref.mpjpe(p3, r3) <= 0.05 and float((p3 - r3).abs().max()) <= 0.5
