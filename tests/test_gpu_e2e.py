"""GPU end-to-end: the Pose3dEstimator drop-in vs golden vectors of the reference's
_estimate_poses_batched (geometry -> sampler -> backbone -> head -> reconstruction -> TTA post).

This is NOT the 1e-3 mm parity claim (that one is on identical crops / features, see
test_gpu_decode_recon.py and test_gpu_head.py): here crops differ by the sampler's fp32 noise and the
tiny backbone runs on MIOpen instead of oneDNN, and the head amplifies both.  Bounds: per case, ~3x what
was measured on MI355X in round 4 (gpurun_out/r04a_e2e.log; round 3 allowed 0.05 mm MPJPE / 0.5 mm max
everywhere, 10 - 70x the measured values: a sampler regression of 10x passed); poses2d median 2e-3 px /
95th percentile 0.05 px."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import cases, cpu_ref

pytestmark = pytest.mark.gpu

# poses3d from IMAGES vs the stored reference output, (MPJPE, max) in mm; measured (fused head) beside each
E2E_BOUND = {'aug1': (1.5e-2, 6e-2),            # 4.5e-3, 2.0e-2
             'aug5': (1e-2, 2.5e-2),            # 3.0e-3, 7.6e-3
             'aug5_dist_aa2': (5e-3, 2e-2),     # 1.4e-3, 6.1e-3
             'aug4_dist12': (2e-2, 4e-2),       # 6.4e-3, 1.2e-2
             'aug2_aa8': (3e-3, 1.2e-2),        # 7.8e-4, 3.3e-3
             'aug2_aa4_bigbox': (6e-3, 1.2e-2)}  # 1.8e-3, 3.6e-3


def build_estimator(case, fused_head):
    from metrabs_amd.config import MetrabsConfig
    from metrabs_amd.joint_info import JointInfo
    from metrabs_amd.models.metrabs import Metrabs
    from metrabs_amd.multiperson.multiperson_model import Pose3dEstimator
    cfg = MetrabsConfig.from_any(case['cfg'].as_dict())
    ji = JointInfo(cases.COCO17, cases.COCO17_EDGES)
    model = Metrabs(case['backbone'], ji, cfg, in_channels=case.get('C', cases.E2E_C), fused_head=fused_head)
    with torch.no_grad():
        model.heatmap_heads.conv_final.weight.copy_(case['head_w'][:, :, None, None])
        model.heatmap_heads.conv_final.bias.copy_(case['head_b'])
    model = model.cuda().eval()
    skel = {'': dict(indices=case['skeleton'], names=[cases.COCO17[i] for i in case['skeleton']],
                     edges=[[0, 1]])}
    return Pose3dEstimator(model, skel, case['jtm'])


@pytest.mark.parametrize('fused_head', [False, True])
@pytest.mark.parametrize('name', list(cases.E2E_CASES))
def test_estimate_poses_vs_golden(name, fused_head, hip_lib):
    g = load_golden(f'e2e_{name}')
    case = cases.e2e_case(name)
    est = build_estimator(case, fused_head)
    with torch.inference_mode():
        res = est._estimate_poses_batched(
            case['images'], case['boxes'], case['K'], case['dist'], case['extr'], case['world_up'],
            55, case['ibs'], case['aa'], case['num_aug'], case['average_aug'], '', False)
    assert [len(p) for p in res['poses3d']] == g['counts'].tolist()
    p3 = torch.cat(res['poses3d']).cpu()
    p2 = torch.cat(res['poses2d']).cpu()
    g3, g2 = torch.from_numpy(g['poses3d']), torch.from_numpy(g['poses2d'])
    assert p3.shape == g3.shape and p2.shape == g2.shape
    print(f'[parity] e2e {name} fused={fused_head}: poses3d MPJPE {cpu_ref.mpjpe(p3, g3):.2e} mm '
          f'max {float((p3 - g3).abs().max()):.2e} mm; poses2d max {float((p2 - g2).abs().max()):.2e} px')
    b_mpjpe, b_max = E2E_BOUND[name]
    assert cpu_ref.mpjpe(p3, g3) <= b_mpjpe and float((p3 - g3).abs().max()) <= b_max
    # random-weight heads put some joints at near-zero depth where x/z amplifies any difference:
    # bound the bulk of the distribution, not the projection singularities
    d2 = (p2 - g2).abs().flatten()
    assert float(d2.median()) <= 2e-3 and float(torch.quantile(d2, 0.95)) <= 0.05


@pytest.mark.parametrize('dtype', [torch.float32, torch.int16, torch.float16])
def test_frames_of_other_dtypes_are_taken_as_the_reference_takes_them(dtype, hip_lib):
    """The reference never asks for uint8: `(images.float() / 255) ** 2.2` (multiperson_model.py:196).  Frames of
    another dtype holding the same values must give the golden output of the uint8 frames (same bound), through
    the materialised f32 level 0, from the host or the device, with graphs requested or not (such frames stay
    eager)."""
    name = 'aug5'
    g = load_golden(f'e2e_{name}')
    case = cases.e2e_case(name)
    est = build_estimator(case, True)
    est.graph_batches = True
    g3 = torch.from_numpy(g['poses3d'])
    b_mpjpe, b_max = E2E_BOUND[name]
    for frames in (case['images'].to(dtype), case['images'].to(dtype).cuda()):
        for _ in range(2):
            res = est.estimate_poses_batched(
                frames, case['boxes'], intrinsic_matrix=case['K'], distortion_coeffs=case['dist'],
                extrinsic_matrix=case['extr'], world_up_vector=case['world_up'], internal_batch_size=case['ibs'],
                antialias_factor=case['aa'], num_aug=case['num_aug'], average_aug=case['average_aug'])
            p3 = torch.cat(res['poses3d']).cpu()
            assert cpu_ref.mpjpe(p3, g3) <= b_mpjpe and float((p3 - g3).abs().max()) <= b_max
    assert est.graphs.stats['captures'] == 0 and not est.graphs.frame_sets


def test_public_api_shapes_and_detector(hip_lib):
    """detect_poses / estimate_poses (single image) and the batched variants, pluggable detector,
    an image with zero boxes, skeleton selection, average_aug=False."""
    case = cases.e2e_case('aug5')
    est = build_estimator(case, False)
    boxes = case['boxes']

    def detector(images, threshold, nms_iou_threshold, max_detections):
        return boxes

    est.detector = detector
    with torch.inference_mode():
        r = est.detect_poses_batched(case['images'], case['K'], num_aug=2, internal_batch_size=4)
        assert set(r) == {'boxes', 'poses3d', 'poses2d'}
        assert [p.shape for p in r['poses3d']] == [(len(b), 17, 3) for b in boxes]
        assert r['poses3d'][1].shape == (0, 17, 3)  # image without detections
        r1 = est.estimate_poses(case['images'][0], boxes[0][:, :4], case['K'][0], num_aug=3,
                                average_aug=False)
        assert r1['poses3d'].shape == (len(boxes[0]), 3, 17, 3) and 'boxes' not in r1
        assert r1['poses2d'].shape == (len(boxes[0]), 3, 17, 2)
        r2 = est.estimate_poses_batched(case['images'], [b[:, :4] for b in boxes], num_aug=1)
        assert torch.isfinite(torch.cat(r2['poses3d'])).all()
        est.detector = None
        with pytest.raises(RuntimeError):
            est.detect_poses(case['images'][0])


def test_frames_beyond_the_2gib_descriptor_take_per_batch_pyramids(hip_lib, monkeypatch):
    """mtr_warp_crops_u8 addresses a call's uint8 frames through one 32-bit-offset descriptor
    (< 2 GiB); beyond that Pose3dEstimator builds, per internal batch, the pyramid of the frames
    that batch references.  Same kernels on the same texels: bit-equal to the one-pyramid call
    (the threshold is lowered instead of allocating 2 GiB of frames)."""
    from metrabs_amd import kernels
    case = cases.e2e_case('aug5')
    est = build_estimator(case, True)
    args = (case['images'], case['boxes'], case['K'], case['dist'], case['extr'], case['world_up'],
            55, case['ibs'], case['aa'], case['num_aug'], case['average_aug'], '', False)
    with torch.inference_mode():
        ref = est._estimate_poses_batched(*args)
        monkeypatch.setattr(kernels, 'MAX_U8_FRAME_BYTES', 1)
        calls = []
        orig = kernels.build_pyramid
        monkeypatch.setattr(kernels, 'build_pyramid', lambda im, **kw: calls.append(len(im)) or orig(im, **kw))
        with pytest.raises(ValueError):  # one batch's frames are themselves over the (lowered) limit
            est._estimate_poses_batched(*args)
        monkeypatch.setattr(kernels, 'MAX_U8_FRAME_BYTES', case['images'][:1].numel() * 2 + 1)
        calls.clear()
        out = est._estimate_poses_batched(*args)
    if len(case['images']) > 2:
        assert calls and max(calls) <= 2  # a pyramid per internal batch, only its own frames
    for k in ('poses3d', 'poses2d'):
        assert all(torch.equal(a, b) for a, b in zip(ref[k], out[k]))


class _InjectedFeatures(torch.nn.Module):
    """Stands in for the backbone: call i returns the features the REFERENCE's backbone produced in
    its call i of the same estimate (golden e2efeat_*), whatever crops it is handed."""

    def __init__(self, golden, out_channels):
        super().__init__()
        self.feats = [torch.from_numpy(golden[f'features_{i}']) for i in range(int(golden['n_calls']))]
        self.out_channels = out_channels
        self.calls = 0

    def forward(self, crops):
        f = self.feats[self.calls].to(crops.device)
        assert f.shape[0] == crops.shape[0], 'internal batches are split as the reference splits them'
        self.calls += 1
        return f


@pytest.mark.parametrize('fused_head', [True, False])
@pytest.mark.parametrize('name', list(cases.E2E_FEATURE_CASES))
def test_estimator_glue_on_injected_reference_features_within_1e3_mm(name, fused_head, hip_lib):
    """Pose3dEstimator with the sampler's influence removed: the backbone returns the features the
    reference's own backbone produced (stored per crop-model call), so what is compared is the glue --
    box -> crop geometry (new intrinsics, rotations), internal-batch splitting, head, absolute
    reconstruction with its batch-global scalars, mirror un-swap / back-rotation / joint transform /
    world transform / skeleton selection / TTA mean (K7) -- against the poses of that same reference
    run.  Gate: the north star's 1e-3 mm MPJPE (the end-to-end test from images allows 0.05 mm because
    sampler noise x head gain sits in it)."""
    g = load_golden(f'e2efeat_{name}')
    case = cases.e2e_case(name)
    est = build_estimator(case, fused_head)
    inj = _InjectedFeatures(g, cases.E2E_C)
    est.crop_model.backbone = inj
    with torch.inference_mode():
        res = est._estimate_poses_batched(
            case['images'], case['boxes'], case['K'], case['dist'], case['extr'], case['world_up'],
            55, case['ibs'], case['aa'], case['num_aug'], case['average_aug'], '', False)
    assert inj.calls == int(g['n_calls'])
    p3, p2 = torch.cat(res['poses3d']).cpu(), torch.cat(res['poses2d']).cpu()
    g3, g2 = torch.from_numpy(g['poses3d']), torch.from_numpy(g['poses2d'])
    assert p3.shape == g3.shape and p2.shape == g2.shape
    err, mx = cpu_ref.mpjpe(p3, g3), float((p3 - g3).abs().max())
    print(f'[parity] glue {name} fused={fused_head}: poses3d MPJPE {err:.2e} mm max {mx:.2e} mm; '
          f'poses2d max {float((p2 - g2).abs().max()):.2e} px')
    assert err <= 1e-3, (name, err)
    assert mx <= 6e-3, (name, mx)
    # poses2d: the tiny random head puts some joints at near-zero depth, where x/z amplifies any
    # difference without bound (up to 1e6 px in these cases): the bulk of the distribution is gated
    d2 = (p2 - g2).abs().flatten()
    print(f'[parity] glue {name}: poses2d median {float(d2.median()):.1e} px, 75 % {float(torch.quantile(d2, 0.75)):.1e} px')
    assert float(d2.median()) <= 1e-4 and float(torch.quantile(d2, 0.75)) <= 1e-3


@pytest.mark.parametrize('name', list(cases.E2E_CASES))
def test_sampler_contribution_to_the_pose_error_in_mm(name, hip_lib):
    """Attribution: how many mm of the from-images distance are the SAMPLER's?  OUR
    crops (geometry kernel + warp kernel on the GPU) and the ORACLE's crops (cpu_ref.get_crops) go through the
    SAME CPU crop model (the oracle's backbone, head and reconstruction) with the oracle's intrinsics: the two
    results differ by what the crops differ -- sampler arithmetic + the geometry's rounding (ours forms
    inv(K_new R) in f64, the reference in f32) -- and by nothing else (no MIOpen-vs-oneDNN backbone, no head).
    Printed per case; gated at a fixed multiple of what was measured when the test was written."""
    case = cases.e2e_case(name)
    est = build_estimator(case, True)
    ours_crops = []
    est.crop_model.register_forward_pre_hook(lambda m, a: ours_crops.append(a[0][0].detach().float().cpu()))
    est.graph_batches = False
    args = (case['images'], case['boxes'], case['K'], case['dist'], case['extr'], case['world_up'], 55, case['ibs'],
            case['aa'], case['num_aug'], case['average_aug'])
    with torch.inference_mode():
        est._estimate_poses_batched(*args, '', False)
        backbone_cpu = cases.e2e_case(name)['backbone']
        head = lambda crops, K: cpu_ref.crop_model_from_features(backbone_cpu(crops), case['head_w'], case['head_b'], K,
                                                                 17, case['cfg'])
        theirs_crops, calls = [], [0]

        def oracle_model(inp):
            theirs_crops.append(inp[0].clone())
            return head(*inp)

        def swapped_model(inp):   # the oracle's call i, fed OUR crops of internal batch i
            mine = ours_crops[calls[0]].reshape(inp[0].shape)
            calls[0] += 1
            return head(mine, inp[1])

        mm = cases.mirror_mapping(cases.COCO17)
        ref = cpu_ref.estimate_poses_batched(oracle_model, mm, 17, case['res'], *args)
        swp = cpu_ref.estimate_poses_batched(swapped_model, mm, 17, case['res'], *args)
    assert calls[0] == len(ours_crops) == len(theirs_crops)
    a, b = torch.cat(ref['poses3d']), torch.cat(swp['poses3d'])
    dc = torch.cat([(o.reshape(t.shape) - t).abs().flatten() for o, t in zip(ours_crops, theirs_crops)])
    err, mx = cpu_ref.mpjpe(a, b), float((a - b).abs().max())
    print(f'[parity] sampler contribution {name}: crops differ max {float(dc.max()):.2e} mean {float(dc.mean()):.2e} '
          f'(gamma-encoded units) -> poses3d MPJPE {err:.2e} mm, max {mx:.2e} mm through the same CPU crop model')
    assert err <= SAMPLER_MM_BOUND[name][0] and mx <= SAMPLER_MM_BOUND[name][1]


# (MPJPE, max) in mm: ~3x the values measured on MI355X in round 5 (profiles/r05f_sampler_mm.log: 4.3e-3 / 2.2e-2,
# 9.8e-4 / 2.0e-3, 1.5e-3 / 6.8e-3, 2.7e-3 / 5.6e-3, 1.0e-3 / 3.6e-3, 7.1e-3 / 1.9e-2) -- the crops differ by the
# GEOMETRY's rounding (ours inverts K_new R in f64, the reference in f32: tests/test_gpu_sampler.py prints 9.2e-6
# mean in linear light) more than by the sampling arithmetic (3.7e-6 ours, 4.1e-6 the reference), times the head's gain
SAMPLER_MM_BOUND = {'aug1': (1.3e-2, 6.5e-2), 'aug5': (3e-3, 6e-3), 'aug5_dist_aa2': (4.5e-3, 2e-2),
                    'aug4_dist12': (8e-3, 1.7e-2), 'aug2_aa8': (3e-3, 1.1e-2), 'aug2_aa4_bigbox': (2.1e-2, 5.6e-2)}
