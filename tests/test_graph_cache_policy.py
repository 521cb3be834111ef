"""CPU: the policy of the estimator's HIP-graph cache (metrabs_amd/graph_cache.py) with the GPU objects
stubbed -- when a shape is captured ('auto': 2nd occurrence, True: 1st), LRU order, the eviction rate limit of
a full cache, graphs dropped with the frame set they read, a failed capture leaving its shape eager."""
import types

import pytest
import torch

from metrabs_amd import graph_cache


class FakeFrames:
    def __init__(self, n, h, w, device, hwc=False):
        self.key = (h, w, str(device), hwc)
        self.capacity = n
        self.hwc = hwc


class FakeGraph:
    fail = False
    stale = False

    def __init__(self, est, frames, batch_args, tta, aa, post, n_frames=None, pool=None):
        if FakeGraph.fail:
            raise RuntimeError('capture failed')
        self.frames = frames
        self.n_frames = n_frames

    def is_current(self, est):
        return not FakeGraph.stale


@pytest.fixture
def cache(monkeypatch):
    monkeypatch.setattr(graph_cache, 'FrameSet', FakeFrames)
    monkeypatch.setattr(graph_cache, 'BatchGraph', FakeGraph)
    monkeypatch.setattr(torch.cuda, 'is_current_stream_capturing', lambda: False)
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
    FakeGraph.fail = FakeGraph.stale = False
    est = types.SimpleNamespace(graph_batches='auto', crop_dtype=torch.float32, crop_channels_last=False,
                                crop_model=types.SimpleNamespace(input_resolution=256),
                                _device=lambda: torch.device('cpu'))
    return graph_cache.GraphCache(est, max_graphs=2, max_frame_sets=1, min_batches_between_evictions=5)


TTA = dict(gammas=torch.ones(1))
POST = dict(joint_transform=None, average_aug=True, skeleton=torch.arange(17))


def call(cache, n_boxes, n_frames=2, hw=8, interleaved=False):
    """One call with a single internal batch of n_boxes -> 'replay' | 'capture' | 'eager'."""
    images = torch.zeros(n_frames, 3, hw, hw, dtype=torch.uint8)
    if interleaved:
        images = images.contiguous(memory_format=torch.channels_last)
    before = dict(cache.stats)
    plan = cache.plan_call(images, [(0, n_boxes)], TTA, 1, POST)
    if plan is None:
        return 'eager'
    g = plan.graph_for(0, ())
    if g is None:
        return 'eager'
    return 'capture' if cache.stats['captures'] > before['captures'] else 'replay'


def test_auto_captures_on_the_second_occurrence_and_true_on_the_first(cache):
    assert [call(cache, 4) for _ in range(3)] == ['eager', 'capture', 'replay']
    cache.est.graph_batches = True
    assert [call(cache, 5) for _ in range(2)] == ['capture', 'replay']
    assert cache.stats['captures'] == 2 and cache.stats['replays'] == 2


def test_full_cache_evicts_at_most_once_per_interval(cache):
    cache.est.graph_batches = True
    assert call(cache, 1) == 'capture' and call(cache, 2) == 'capture'   # full (2 graphs)
    assert call(cache, 3) == 'capture'            # the first eviction is free (LRU = the 1-box graph)
    assert call(cache, 1) == 'eager'              # evicted, and no second eviction within 5 batches
    assert call(cache, 2) == 'replay' and call(cache, 3) == 'replay'
    assert call(cache, 4) == 'eager'
    outcomes = [call(cache, 4) for _ in range(3)]  # batches 8, 9, 10 of the cache: the interval has passed
    assert 'capture' in outcomes and cache.stats['evictions'] == 2
    assert len(cache.graphs) == 2


def test_one_frame_set_per_frame_size_grows_to_the_largest_frame_count(cache):
    cache.est.graph_batches = True
    assert call(cache, 4, n_frames=3) == 'capture'
    assert call(cache, 4, n_frames=2) == 'capture'     # fewer frames of the same size: the head of the same set
    assert len(cache.frame_sets) == 1 and cache.stats['evictions'] == 0 and len(cache.graphs) == 2
    assert {g.n_frames for g in cache.graphs.values()} == {2, 3}
    assert call(cache, 4, n_frames=3) == 'replay' and call(cache, 4, n_frames=2) == 'replay'
    assert call(cache, 4, n_frames=5) == 'capture'     # more frames than the set holds: a larger set ...
    fs = next(iter(cache.frame_sets.values()))
    assert fs.capacity == 5 and cache.stats['evictions'] == 2   # ... and the graphs over the old one are gone
    assert all(g.frames is fs for g in cache.graphs.values())


def test_interleaved_frames_get_a_frame_set_and_graphs_of_their_own(cache):
    """Frames over [N,H,W,3] memory stay interleaved in their frame set (the sampler reads them as they
    lie): the layout is part of the frame-set key and of every graph key."""
    cache.est.graph_batches = True
    cache.max_frame_sets = 2
    assert call(cache, 4) == 'capture' and call(cache, 4, interleaved=True) == 'capture'
    assert len(cache.frame_sets) == 2 and sorted(fs.hwc for fs in cache.frame_sets.values()) == [False, True]
    assert call(cache, 4) == 'replay' and call(cache, 4, interleaved=True) == 'replay'
    assert cache.stats['evictions'] == 0


def test_cycling_over_more_frame_sizes_than_sets_does_not_recapture_every_call(cache):
    """ max_frame_sets = 1: the second frame size may replace the first once per eviction interval;
    in between it runs eagerly, and the size that owns the set keeps replaying."""
    cache.est.graph_batches = True
    assert call(cache, 4, hw=8) == 'capture'
    assert call(cache, 4, hw=16) == 'capture'          # the first replacement is free
    outcomes = [call(cache, 4, hw=hw) for hw in (8, 16, 8, 16)]
    assert outcomes == ['eager', 'replay', 'eager', 'replay'], outcomes
    assert cache.stats['captures'] == 2
    # the pinned-frame staging of an eager call does not replace the set inside the interval either
    assert cache.frame_set(2, 8, 8, torch.device('cpu'), optional=True) is None
    # ... once the interval (5 batches) has passed a size that keeps coming replaces it
    assert 'capture' in [call(cache, 4, hw=8) for _ in range(3)]
    assert cache.stats['captures'] == 3 and len(cache.frame_sets) == 1


def test_a_graph_whose_weights_were_replaced_is_captured_again(cache):
    cache.est.graph_batches = True
    assert call(cache, 4) == 'capture' and call(cache, 4) == 'replay'
    FakeGraph.stale = True
    try:
        assert call(cache, 4) == 'capture' and cache.stats['stale'] == 1
    finally:
        FakeGraph.stale = False
    assert call(cache, 4) == 'replay'


def test_a_failed_capture_leaves_the_shape_eager(cache):
    cache.est.graph_batches = True
    FakeGraph.fail = True
    with pytest.warns(UserWarning):
        assert call(cache, 4) == 'eager'
    FakeGraph.fail = False
    assert call(cache, 4) == 'eager' and cache.last_capture_error   # not retried
    assert call(cache, 5) == 'capture'                               # other shapes are


def test_empty_ranges_and_disabled_cache(cache):
    images = torch.zeros(2, 3, 8, 8, dtype=torch.uint8)
    assert cache.plan_call(images, [(3, 3)], TTA, 1, POST) is None    # an empty slice is never a graph
    assert cache.stats['captures'] == 0


def test_eager_staging_advances_the_eviction_interval_on_its_own(cache):
    """ graph_batches off + pinned host frames: only frame_set(optional=True) runs.  The first
    replacement must not block every later one: each staging call counts as a batch of the interval."""
    dev = torch.device('cpu')
    a = cache.frame_set(2, 8, 8, dev, optional=True)
    assert a is not None
    b = cache.frame_set(2, 16, 16, dev, optional=True)      # the first replacement is free
    assert b is not None and len(cache.frame_sets) == 1
    got = [cache.frame_set(2, 8, 8, dev, optional=True) for _ in range(6)]
    assert got[0] is None                                    # inside the interval: plain upload
    assert any(g is not None for g in got)                   # ... and allowed again once 5 staging calls went by


def test_eager_staging_never_drops_a_frame_set_that_live_graphs_read(cache):
    """ an eager call replays no graph: its staging takes the plain upload rather than replacing (or
    growing) a frame set captured graphs read."""
    cache.est.graph_batches = True
    assert call(cache, 4, hw=8) == 'capture'
    dev = torch.device('cpu')
    for _ in range(8):                                       # (well past the eviction interval)
        assert cache.frame_set(2, 16, 16, dev, optional=True) is None     # would push out the 8x8 set (LRU)
        assert cache.frame_set(5, 8, 8, dev, optional=True) is None       # would grow, i.e. replace, the 8x8 set
    assert len(cache.graphs) == 1 and cache.stats['evictions'] == 0
    assert cache.frame_set(2, 8, 8, dev, optional=True) is next(iter(cache.frame_sets.values()))
    assert call(cache, 4, hw=8) == 'replay'


def test_backbone_fingerprint_sees_the_backbone_alone_being_cast_moved_or_replaced():
    """ Metrabs._apply only sees .half() / .to() on the WHOLE crop model; a graph must also go stale when
    the backbone alone changes storage."""
    bb = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.ReLU(), torch.nn.Sequential(torch.nn.Conv2d(4, 4, 1), torch.nn.ReLU()))
    model = types.SimpleNamespace(backbone=bb)
    fp = graph_cache.backbone_fingerprint(model)
    assert fp == graph_cache.backbone_fingerprint(model)
    assert graph_cache._last_parameter(bb) is list(bb.parameters())[-1]
    bb.half()
    fp16 = graph_cache.backbone_fingerprint(model)
    assert fp16 != fp
    bb[2][0] = torch.nn.Conv2d(4, 4, 1).half()              # the last layer replaced
    assert graph_cache.backbone_fingerprint(model) != fp16
    model.backbone = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1))
    assert graph_cache.backbone_fingerprint(model)[0] != fp[0]
    assert graph_cache.backbone_fingerprint(types.SimpleNamespace(backbone=lambda x: x))   # (a callable stub: its identity)
