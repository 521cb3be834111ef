"""GPU parity of K8, the plausibility filter + pose NMS behind the hot path (row f.2), through the
C-ABI: kept indices and validity masks are integers / booleans -> exact equality with the oracle
(oracle/cpu_ref.py:filter_poses, itself pinned to the reference functions that run,
tests/test_oracle_pin.py::test_pose_filter_vs_golden) on cases whose decisions sit away from the
thresholds (oracle/cases.py:filter_case reports the margins)."""
import pytest
import torch

from conftest import load_golden
from oracle import cases, cpu_ref

pytestmark = pytest.mark.gpu


def run_kernel(c, unbiased=False, order='index'):
    from metrabs_amd import kernels
    counts = [len(b) for b in c['boxes']]
    p3, p2, bx = torch.cat(c['poses3d']).cuda(), torch.cat(c['poses2d']).cuda(), torch.cat(c['boxes']).cuda()
    keep_idx, keep_count, valid = kernels.filter_poses(
        p3, p2, bx, counts, c['edges'], c['mean_bones'], n_joints=c['n_joints'], unbiased=unbiased, order=order)
    keep_idx, keep_count, valid = keep_idx.cpu(), keep_count.cpu().tolist(), valid.cpu()
    out, masks, start = [], [], 0
    for n, k in zip(counts, keep_count):
        out.append((keep_idx[start:start + k] - start).long())
        assert bool((keep_idx[start + k:start + n] == -1).all())
        masks.append(valid[start:start + n])
        start += n
    return out, masks


@pytest.mark.parametrize('order', ['index', 'score'])
@pytest.mark.parametrize('name', list(cases.FILTER_CASES))
def test_filter_vs_oracle_and_golden(name, order, hip_lib):
    c = cases.filter_case(name)
    want, want_masks = cpu_ref.filter_poses(c['boxes'], c['poses3d'], c['poses2d'], c['edges'],
                                            c['mean_bones'], order=order)
    got, masks = run_kernel(c, order=order)
    g = load_golden(f'filter_{name}')
    for i, (a, b, ma, mb) in enumerate(zip(got, want, masks, want_masks)):
        assert torch.equal(ma, mb), (name, i, ma, mb)
        assert a.tolist() == b.tolist(), (name, i, a, b, c['kinds'][i])
        if order == 'index' and len(c['boxes'][i]):
            assert a.tolist() == g[f'keep_{i}'].tolist()  # what the reference's NMS returns for this mask
    kept_kinds = [c['kinds'][i][j] for i in range(len(got)) for j in got[i].tolist()]
    assert kept_kinds and all(k == 'person' for k in kept_kinds), kept_kinds


@pytest.mark.parametrize('name', ['coco17_aug5', 'chain40_aug3'])
def test_unbiased_variance_option_matches_the_pytorch_port(name, hip_lib):
    c = cases.filter_case(name)
    want, want_masks = cpu_ref.filter_poses(c['boxes'], c['poses3d'], c['poses2d'], c['edges'],
                                            c['mean_bones'], unbiased=True)
    got, masks = run_kernel(c, unbiased=True)
    for a, b, ma, mb in zip(got, want, masks, want_masks):
        assert torch.equal(ma, mb) and a.tolist() == b.tolist()


def test_ragged_and_degenerate_cases(hip_lib):
    """No images with poses, one pose, J < 4 (k = 0: the reference's mean over an empty top-k is NaN
    -> nothing is suppressed), no bone table."""
    from metrabs_amd import kernels
    g = cases.gen(5)
    # J = 3 -> k = 0: identical poses must both survive
    p3 = (torch.randn(1, 1, 3, 3, generator=g) * 200 + torch.tensor([0.0, 0.0, 3000.0])).repeat(2, 2, 1, 1)
    p2 = 1000 * p3[..., :2] / p3[..., 2:] + 500
    lo, hi = p2.mean(1).min(1).values, p2.mean(1).max(1).values
    bx = torch.cat([lo - 5, hi - lo + 10, torch.tensor([[0.9], [0.8]])], dim=1)
    keep_idx, keep_count, valid = kernels.filter_poses(p3.cuda(), p2.cuda(), bx.cuda(), [2, 0], None, None)
    assert keep_count.tolist() == [2, 0] and keep_idx.tolist() == [0, 1] and valid.tolist() == [True, True]
    # exact duplicates at J = 17: the lower score goes
    c = cases.filter_case('coco17_aug1')
    p3 = c['poses3d'][0][:1].repeat(2, 1, 1, 1)
    p2 = c['poses2d'][0][:1].repeat(2, 1, 1, 1)
    bx = c['boxes'][0][:1].repeat(2, 1)
    bx[:, 4] = torch.tensor([0.4, 0.7])
    keep_idx, keep_count, valid = kernels.filter_poses(p3.cuda(), p2.cuda(), bx.cuda(), [2], None, None)
    assert keep_count.tolist() == [1] and keep_idx.tolist() == [1, -1]


def test_box_consistency_known_answer(hip_lib):
    """is_pose_consistent_with_box (TF plausibility_check.py:66-84): the hand-derived cases of
    cases.box_consistency_kat through K8 (no bone table, one augmentation: the box test alone
    decides `valid`), incl. the strict '>' at exactly half the box area."""
    from metrabs_amd import kernels
    pose2d, boxes, want = cases.box_consistency_kat()
    n = len(boxes)
    p2 = pose2d[:, None]                                            # [n, A=1, J=3, 2]
    p3 = torch.cat([p2, torch.full_like(p2[..., :1], 3000.0)], dim=-1) + \
        torch.arange(n).reshape(n, 1, 1, 1) * 5000.0                # far apart: the NMS keeps all
    keep_idx, keep_count, valid = kernels.filter_poses(p3.cuda(), p2.contiguous().cuda(), boxes.cuda(),
                                                       [n], None, None)
    assert valid.cpu().tolist() == want.tolist()
    assert keep_count.tolist() == [int(want.sum())]


def test_detect_poses_applies_the_filter_when_bone_lengths_are_set(hip_lib):
    """Pose3dEstimator.detect_poses_batched(suppress_implausible_poses=True): ignored without bone
    lengths (the PyTorch reference's behaviour), K8 with them -- same poses, a subset of the rows."""
    from test_gpu_e2e import build_estimator
    case = cases.e2e_case('aug5')
    est = build_estimator(case, fused_head=True)
    boxes = [b.clone() for b in case['boxes']]
    boxes[0] = torch.cat([boxes[0], boxes[0][:1] + torch.tensor([2.0, -1.0, 1.0, 0.5, -0.3])])  # a duplicate box
    est.detector = lambda images, threshold, nms_iou_threshold, max_detections: boxes
    args = (case['images'], case['K'], case['dist'], case['extr'], case['world_up'], 55, case['ibs'], case['aa'],
            case['num_aug'], False, '', 0.3, 0.7, 150, False, True)
    with torch.inference_mode():
        plain = est.detect_poses_batched(*args)
        est.mean_bone_lengths = torch.full((len(cases.COCO17_EDGES),), 1e9)  # every bone "too short": rel < 0.1
        none_left = est.detect_poses_batched(*args)
        est.mean_bone_lengths = None
        est.mean_bone_lengths = torch.full((len(cases.COCO17_EDGES),), 250.0)
        filtered = est.detect_poses_batched(*args)
    assert [len(p) for p in plain['poses3d']] == [len(b) for b in boxes]
    assert all(len(p) == 0 for p in none_left['poses3d'])
    for pf, pp, bf in zip(filtered['poses3d'], plain['poses3d'], filtered['boxes']):
        assert len(pf) <= len(pp) and len(bf) == len(pf)
        for row in pf:  # every surviving pose is one of the unfiltered ones, bit for bit
            assert bool((pp == row).flatten(1).all(1).any())


@pytest.mark.parametrize('A,J,n', [(10, 17, 6), (3, 122, 20), (2, 40, 130), (1, 6, 3)])
def test_filter_sizes_beyond_the_common_ones(A, J, n, hip_lib):
    """More augmentations than the 8 kept in registers, multi-skeleton joint counts, > 128 poses in
    one image (several rounds of the 4-wave pose loop), tiny skeletons: vs the oracle on noise-free
    far-apart people plus exact duplicates (decisions far from the thresholds)."""
    g = cases.gen(600 + A + J + n)
    edges = [(i, i + 1) for i in range(J - 1)]
    template = torch.cumsum(torch.randn(J, 3, generator=g) * torch.tensor([60.0, 90.0, 40.0]), dim=0)
    tj = torch.tensor(edges)
    mean_bones = torch.norm(template[tj[:, 0]] - template[tj[:, 1]], dim=-1)
    people = []
    for i in range(n):
        base = template + torch.tensor([(i % 16) * 2500.0 - 20000.0, (i // 16) * 3000.0, 9000.0])
        if i % 5 == 4:  # an exact duplicate of the previous person, lower score
            base = people[-1][0][0].clone()
        people.append((base[None].repeat(A, 1, 1) + torch.randn(A, J, 3, generator=g) * 2.0, 0.9 - 0.001 * i))
    p3 = torch.stack([p for p, _ in people])
    p2 = 1500.0 * p3[..., :2] / p3[..., 2:] + torch.tensor([960.0, 540.0])
    m2 = p2.mean(1)
    lo, hi = m2.min(1).values, m2.max(1).values
    bx = torch.cat([lo - 10, hi - lo + 20, torch.tensor([[s] for _, s in people])], dim=1)
    c = dict(boxes=[bx], poses3d=[p3], poses2d=[p2], edges=edges, mean_bones=mean_bones, n_joints=J)
    want, want_masks = cpu_ref.filter_poses(c['boxes'], c['poses3d'], c['poses2d'], edges, mean_bones)
    got, masks = run_kernel(c)
    assert torch.equal(masks[0], want_masks[0])
    assert got[0].tolist() == want[0].tolist()
    if J >= 17:  # (the 6-joint chain is too narrow for its own box test: all rejected, on both sides)
        assert len(got[0]) == n - n // 5  # every fifth pose is a duplicate
