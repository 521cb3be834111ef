"""HIP graphs behind the drop-in API (metrabs_amd/graph_cache.py): Pose3dEstimator.estimate_poses_batched
with graph_batches on must return the eager path's bits -- across different frames, boxes and cameras, with
several internal batches per call, ragged tails, host-resident frames, and frame-set eviction -- and it
must actually replay graphs (stats), not silently stay eager."""
import numpy as np
import pytest
import torch

from oracle import cases
from test_gpu_e2e import build_estimator

pytestmark = pytest.mark.gpu


def _inputs(case, seed, n_images, per_image):
    """Fresh frames / boxes / cameras of the case's frame size."""
    g = torch.Generator().manual_seed(seed)
    _, _, h, w = case['images'].shape
    images = torch.randint(0, 256, (n_images, 3, h, w), dtype=torch.uint8, generator=g)
    boxes = []
    for n in per_image:
        xy = torch.rand(n, 2, generator=g) * torch.tensor([w * 0.5, h * 0.5])
        wh = 20 + torch.rand(n, 2, generator=g) * torch.tensor([w * 0.4, h * 0.4])
        boxes.append(torch.cat([xy, wh], dim=1))
    K = torch.stack([cases.intrinsics_for(h, w, 50.0 + 5 * i, seed + i) for i in range(n_images)])
    return images, boxes, K


def _call(est, images, boxes, K, case, **kw):
    args = dict(intrinsic_matrix=K, distortion_coeffs=case['dist'], extrinsic_matrix=case['extr'],
                world_up_vector=case['world_up'], internal_batch_size=case['ibs'], antialias_factor=case['aa'],
                num_aug=case['num_aug'], average_aug=case['average_aug'])
    args.update(kw)
    r = est.estimate_poses_batched(images, boxes, **args)
    return torch.cat(r['poses3d']).clone(), torch.cat(r['poses2d']).clone()


@pytest.mark.parametrize('name', ['aug5', 'aug4_dist12', 'aug5_dist_aa2'])
def test_graphed_api_equals_eager_bit_for_bit(name, hip_lib):
    case = cases.e2e_case(name)
    eager, graphed = build_estimator(case, 'auto'), build_estimator(case, 'auto')
    eager.graph_batches = False
    graphed.graph_batches = True   # capture a shape on its first occurrence
    per_image = [3, 0, 4, 2]       # 9 boxes: at ibs // num_aug boxes per batch, full batches + a ragged tail
    for seed in (11, 12, 13):      # three different inputs through the SAME captured graphs
        images, boxes, K = _inputs(case, seed, 4, per_image)
        a3, a2 = _call(eager, images.cuda(), boxes, K, case)
        b3, b2 = _call(graphed, images.cuda(), boxes, K, case)
        assert torch.isfinite(a3).all() and a3.abs().max() > 0
        assert torch.equal(a3, b3) and torch.equal(a2, b2), (name, seed, float((a3 - b3).abs().max()))
    st = graphed.graphs.stats
    assert st['captures'] >= 1 and st['replays'] >= 2 * st['captures'], st
    assert eager.graphs.stats['captures'] == 0
    # host-resident frames land in the static frame buffer directly: same bits again
    images, boxes, K = _inputs(case, 14, 4, per_image)
    a3, _ = _call(eager, images, boxes, K, case)
    b3, _ = _call(graphed, images.numpy(), boxes, K, case)
    assert torch.equal(a3, b3)
    # pinned host frames: copied on the copy stream through two staging buffers (both paths); the caller's
    # buffer may be overwritten as soon as the call returns
    pinned = images.clone().pin_memory()
    for est in (eager, graphed):
        for seed in (15, 16):
            fresh, boxes, K = _inputs(case, seed, 4, per_image)
            pinned.copy_(fresh)
            want = _call(eager, fresh.cuda(), boxes, K, case)[0]
            got = _call(est, pinned, boxes, K, case)[0]
            pinned.zero_()   # (the call has returned: its copy must not see this)
            assert torch.equal(want, got)
    # another camera set-up / skeleton-free call with other options: new key, still equal
    a3, a2 = _call(eager, images.cuda(), boxes, K, case, average_aug=not case['average_aug'])
    b3, b2 = _call(graphed, images.cuda(), boxes, K, case, average_aug=not case['average_aug'])
    assert torch.equal(a3, b3) and torch.equal(a2, b2)


def test_interleaved_frames_through_the_api(hip_lib):
    """Frames handed over as [N,3,H,W] views of HWC memory (``torch.from_numpy(frames_hwc).permute(0, 3, 1, 2)``,
    channels_last) are sampled in place -- eager, captured, from the device, from pageable and from pinned host
    memory -- with the planar frames' bits, and keep a frame set and graphs of their own."""
    case = cases.e2e_case('aug5')
    eager, graphed = build_estimator(case, 'auto'), build_estimator(case, 'auto')
    eager.graph_batches, graphed.graph_batches = False, True
    per_image = [3, 0, 4, 2]
    for seed in (31, 32, 33):
        images, boxes, K = _inputs(case, seed, 4, per_image)
        hwc = images.permute(0, 2, 3, 1).contiguous()          # what a decoder hands over
        inter = hwc.permute(0, 3, 1, 2)                        # the [N,3,H,W] view of it
        want3, want2 = _call(eager, images.cuda(), boxes, K, case)
        for est in (eager, graphed):
            for frames in (inter.cuda(), inter, torch.from_numpy(hwc.numpy()).permute(0, 3, 1, 2),
                           inter.pin_memory() if seed == 33 else inter):
                got3, got2 = _call(est, frames, boxes, K, case)
                assert torch.equal(want3, got3) and torch.equal(want2, got2), (seed, float((want3 - got3).abs().max()))
    assert all(fs.hwc and not fs.images.is_contiguous() for fs in graphed.graphs.frame_sets.values())
    st = graphed.graphs.stats
    assert st['captures'] >= 1 and st['replays'] >= 4 * st['captures'], st
    # the planar layout of the same frame size: another frame set, other graphs, the same bits
    images, boxes, K = _inputs(case, 34, 4, per_image)
    a3, _ = _call(graphed, images.cuda(), boxes, K, case)
    b3, _ = _call(graphed, images.cuda().contiguous(memory_format=torch.channels_last), boxes, K, case)
    assert torch.equal(a3, b3) and len(graphed.graphs.frame_sets) == 2


def test_auto_mode_captures_on_the_second_occurrence_and_results_do_not_alias(hip_lib):
    case = cases.e2e_case('aug5')
    est = build_estimator(case, 'auto')
    assert est.graph_batches == 'auto'
    images, boxes, K = _inputs(case, 21, 2, [2, 2])   # one internal batch of 4 > ibs // num_aug = 2 -> 2 batches
    r1 = _call(est, images.cuda(), boxes, K, case, internal_batch_size=20)   # 4 boxes: one batch, first sight
    assert est.graphs.stats['captures'] == 0
    r2 = _call(est, images.cuda(), boxes, K, case, internal_batch_size=20)   # second sight: captured + replayed
    assert est.graphs.stats['captures'] == 1
    keep = r2[0].clone()
    images2, boxes2, K2 = _inputs(case, 22, 2, [2, 2])
    r3 = _call(est, images2.cuda(), boxes2, K2, case, internal_batch_size=20)
    assert est.graphs.stats['captures'] == 1 and est.graphs.stats['replays'] >= 1
    assert torch.equal(r1[0], r2[0]) and torch.equal(r2[0], keep), 'a later replay overwrote a returned result'
    assert not torch.equal(r3[0], r2[0])


def test_calls_under_inference_mode_and_plain_no_grad_share_the_cache(hip_lib):
    """The static buffers of the cache outlive the call that made them: a first call under
    torch.inference_mode() must not leave inference tensors that a later call outside it cannot update (and
    the other way round) -- device frames, pinned host frames, captured batches."""
    case = cases.e2e_case('aug5')
    ref = build_estimator(case, 'auto')
    ref.graph_batches = False
    for first_inference in (True, False):
        est = build_estimator(case, 'auto')
        est.graph_batches = True
        for i, seed in enumerate((51, 52, 53, 54)):
            images, boxes, K = _inputs(case, seed, 2, [2, 2])
            frames = images.pin_memory() if i >= 2 else images.cuda()
            want = _call(ref, images.cuda(), boxes, K, case)[0]
            if (i % 2 == 0) == first_inference:
                with torch.inference_mode():
                    got = _call(est, frames, boxes, K, case)[0]
            else:
                got = _call(est, frames, boxes, K, case)[0]
            assert torch.equal(want, got), (first_inference, i)
        assert est.graphs.stats['replays'] >= 2, est.graphs.last_capture_error


def test_frame_set_grows_and_drops_the_graphs_that_read_the_old_one(hip_lib):
    case = cases.e2e_case('aug5')
    ref = build_estimator(case, 'auto')
    ref.graph_batches = False
    est = build_estimator(case, 'auto')
    est.graph_batches = True
    est.graphs.max_frame_sets = 1
    for n_images in (2, 3, 2, 3):  # the 2-frame set is replaced by a 3-frame one, whose head serves 2 frames
        images, boxes, K = _inputs(case, 30 + n_images, n_images, [2] * n_images)
        a = _call(ref, images.cuda(), boxes, K, case)
        b = _call(est, images.cuda(), boxes, K, case)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    fs = next(iter(est.graphs.frame_sets.values()))
    assert est.graphs.stats['evictions'] >= 1 and len(est.graphs.frame_sets) == 1 and fs.capacity == 3
    assert all(g.frames is fs for g in est.graphs.graphs.values())
    assert est.graphs.stats['replays'] >= 1, est.graphs.last_capture_error


def test_graphs_follow_the_heads_weights(hip_lib):
    """ a graph reads the head's packed weights by address: an f32 and an f16 graph live side by
    side (two slots, both kept alive), and an in-place edit of the head's parameters or `.half()` /
    `.float()` on the crop model alone re-captures instead of replaying stale or freed memory."""
    case = cases.e2e_case('aug5')
    est = build_estimator(case, 'auto')
    est.graph_batches = True
    ref = build_estimator(case, 'auto')
    ref.graph_batches = False
    images, boxes, K = _inputs(case, 77, 2, [3, 2])
    want32 = _call(ref, images.cuda(), boxes, K, case)
    got = _call(est, images.cuda(), boxes, K, case)
    assert torch.equal(got[0], want32[0])
    heads = est.crop_model.heatmap_heads
    assert any(slot[0] == 'packed' for slot in heads._derived)
    # in-place edit of the head's bias: the next call must see it
    with torch.no_grad():
        heads.conv_final.bias.add_(0.25)
        ref.crop_model.heatmap_heads.conv_final.bias.add_(0.25)
    want = _call(ref, images.cuda(), boxes, K, case)
    got = _call(est, images.cuda(), boxes, K, case)
    assert torch.equal(got[0], want[0]) and not torch.equal(got[0], want32[0])
    assert est.graphs.stats['stale'] >= 1
    # moving / casting the crop model alone (not through the estimator's _apply)
    stale0 = est.graphs.stats['stale']
    est.crop_model.double().float()
    got = _call(est, images.cuda(), boxes, K, case)
    assert torch.equal(got[0], want[0]) and est.graphs.stats['stale'] > stale0
    got = _call(est, images.cuda(), boxes, K, case)
    assert torch.equal(got[0], want[0]) and est.graphs.stats['replays'] >= 1


def test_loading_weights_drops_the_captured_graphs(hip_lib):
    case = cases.e2e_case('aug5')
    est = build_estimator(case, 'auto')
    est.graph_batches = True
    images, boxes, K = _inputs(case, 70, 2, [2, 2])
    a = _call(est, images.cuda(), boxes, K, case)[0]
    assert len(est.graphs.graphs) >= 1
    sd = {k: v.clone() for k, v in est.crop_model.state_dict().items()}
    sd['heatmap_heads.conv_final.bias'] += 0.5
    est.crop_model.load_state_dict(sd)
    assert len(est.graphs.graphs) == 0            # a replay would still see the old bias
    b = _call(est, images.cuda(), boxes, K, case)[0]
    assert not torch.equal(a, b)
    est.cuda()                                     # nn.Module._apply: dropped as well
    assert len(est.graphs.graphs) == 0


def test_a_full_cache_does_not_capture_on_every_call(hip_lib):
    """More shapes than the cache holds: once full, the cache keeps replaying what it has and runs the other
    shapes eagerly -- at most one eviction per `min_batches_between_evictions` batches."""
    case = cases.e2e_case('aug5')
    ref = build_estimator(case, 'auto')
    ref.graph_batches = False
    est = build_estimator(case, 'auto')
    est.graph_batches = True
    est.graphs.max_graphs = 2
    est.graphs.min_batches_between_evictions = 1000
    for rnd in range(3):
        for n in (1, 2, 3, 4):   # four batch sizes (one internal batch each), a cache of two
            images, boxes, K = _inputs(case, 60 + 10 * rnd + n, 2, [n, 0])
            a = _call(ref, images.cuda(), boxes, K, case, internal_batch_size=100)
            b = _call(est, images.cuda(), boxes, K, case, internal_batch_size=100)
            assert torch.equal(a[0], b[0])
    st = est.graphs.stats
    # the first eviction is allowed at once (the counter starts "long ago"), then none for 1000 batches
    assert st['captures'] <= 3 and len(est.graphs.graphs) == 2 and st['eager_batches'] >= 4, st


def test_empty_call_and_images_without_boxes(hip_lib):
    case = cases.e2e_case('aug5')
    est = build_estimator(case, 'auto')
    est.graph_batches = True
    images, _, K = _inputs(case, 40, 2, [0, 0])
    r = est.estimate_poses_batched(images.cuda(), [torch.zeros(0, 4), torch.zeros(0, 4)], intrinsic_matrix=K,
                                   num_aug=2)
    assert [tuple(p.shape) for p in r['poses3d']] == [(0, 17, 3), (0, 17, 3)]
    assert np.all([tuple(p.shape) == (0, 17, 2) for p in r['poses2d']])
