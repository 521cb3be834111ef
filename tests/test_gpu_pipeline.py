"""GraphedCropPipeline (metrabs_amd/pipeline.py): one internal batch of the hot path captured in a HIP
graph -- what bench.py times.  The replayed graph must give the eager path's numbers, follow its
static input buffers, and run an `after_step` hook (the place of a sharded job's RCCL all-gather)
inside the captured region."""
import pytest
import torch

from oracle import cases
from test_gpu_e2e import build_estimator

pytestmark = pytest.mark.gpu


def make_pipeline(use_graph, num_aug=2):
    from metrabs_amd.pipeline import GraphedCropPipeline
    case = cases.e2e_case('aug5')
    est = build_estimator(case, True)
    n_box = 3
    pipe = GraphedCropPipeline(est, len(case['images']), case['images'].shape[2], case['images'].shape[3], n_box,
                               num_aug=num_aug, use_graph=use_graph)
    pipe.images.copy_(case['images'])
    boxes = torch.cat(case['boxes'])[:n_box, :4]
    pipe.boxes.copy_(boxes)
    pipe.intrinsics.copy_(case['K'][0].expand(n_box, 3, 3))
    pipe.image_ids.copy_(torch.tensor([0, 2, 2], dtype=torch.int32))
    return pipe


def test_graph_replay_equals_eager_and_follows_its_inputs(hip_lib):
    eager, graphed = make_pipeline(False), make_pipeline(True)
    eager.capture()
    graphed.capture()
    a, b = eager.run().clone(), graphed.run().clone()
    assert torch.isfinite(a).all() and torch.equal(a, b)
    # new boxes in the static buffer: the replay (no re-capture) follows
    for p in (eager, graphed):
        p.boxes[:, 0] += 3.0
    a2, b2 = eager.run().clone(), graphed.run().clone()
    assert torch.equal(a2, b2) and not torch.equal(a, a2)


def test_after_step_hook_runs_inside_the_graph(hip_lib):
    pipe = make_pipeline(True)
    seen = torch.zeros(3, 17, 3, device='cuda')
    calls = []

    def hook(poses):
        calls.append(1)
        seen.copy_(poses)

    pipe.after_step = hook
    pipe.capture()
    n_capture_calls = len(calls)   # eager warm-up steps + the capture itself
    seen.zero_()
    out = pipe.run()
    torch.cuda.synchronize()
    assert len(calls) == n_capture_calls, 'a replay runs the captured copy, not the Python hook'
    assert torch.equal(seen, out) and float(out.abs().max()) > 0
