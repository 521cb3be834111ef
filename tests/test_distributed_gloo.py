"""CPU, world_size 2 over gloo: the N>1 path -- whole internal batches dealt round-robin to ranks,
no data-path collective, one all-gather of the poses at the end, optional all-reduce of the three
reconstruction moments (exact-monolithic mode)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_boxes, per_batch, q, exact=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from metrabs_amd import distributed
    r, w, _ = distributed.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    if exact:  # every internal batch cut into one slice per rank (exact-monolithic mode)
        by_rank = distributed.split_internal_batches(n_boxes, per_batch, world)
        ranges = [(a, b) for a, b in by_rank[rank] if b > a]
    else:
        by_rank = [distributed.shard_internal_batches(n_boxes, per_batch, x, world) for x in range(world)]
        ranges = by_rank[rank]
    # a stand-in "crop model": pose of box i = i + small function of its internal batch
    local = torch.cat([torch.arange(a, b, dtype=torch.float32).reshape(-1, 1, 1, 1).repeat(1, 2, 17, 3)
                       + 0.001 * (a // per_batch) for a, b in ranges]) if ranges else \
        torch.zeros(0, 2, 17, 3)
    full = distributed.gather_ranges(local, by_rank, n_boxes)
    if not exact:
        assert torch.equal(full, distributed.gather_poses(local, ranges, n_boxes, per_batch, world))
    moments = torch.tensor([1.0 + rank, 10.0 * (rank + 1), float(len(local))], dtype=torch.float64)
    moments = distributed.allreduce_moments(moments)
    # by value (numpy pickles into the pipe): a torch tensor would travel as a shared-memory handle
    # that disappears if this process exits before the parent has opened it
    q.put((rank, full.numpy(), moments.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_boxes,per_batch,exact', [(23, 4, False), (8, 4, False), (3, 12, False),
                                                      (5, 1, False), (23, 4, True), (5, 1, True), (7, 64, True)])
def test_round_robin_shards_and_single_gather(n_boxes, per_batch, exact):
    world = 2
    ctx = mp.get_context('spawn')
    for attempt in range(3):  # (the probed port can be taken between the probe and the rendezvous)
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, n_boxes, per_batch, q, exact))
                 for r in range(world)]
        for p in procs:
            p.start()
        try:
            results = [q.get(timeout=120) for _ in range(world)]
        except Exception:
            results = None
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        if results is not None and all(p.exitcode == 0 for p in procs):
            break
    else:
        pytest.fail('world-size-2 gloo run failed three times')
    expected = torch.arange(n_boxes, dtype=torch.float32).reshape(-1, 1, 1, 1).repeat(1, 2, 17, 3) \
        + 0.001 * (torch.arange(n_boxes) // per_batch).float().reshape(-1, 1, 1, 1)
    for rank, full, moments in results:
        full, moments = torch.from_numpy(full), torch.from_numpy(moments)
        assert full.shape == (n_boxes, 2, 17, 3)
        assert torch.equal(full, expected), f'rank {rank} gathered a wrong / mis-ordered result'
        assert moments[:2].tolist() == [3.0, 30.0] and moments[2] == n_boxes


def test_exact_monolithic_split_properties():
    """Every internal batch is cut into world_size contiguous slices (one per rank, possibly empty)
    whose sizes differ by at most one box; together they cover every box exactly once."""
    from metrabs_amd.distributed import split_internal_batches
    for n in (0, 1, 7, 64, 65, 256):
        for per in (1, 5, 32, 64):
            for world in (1, 2, 8):
                by_rank = split_internal_batches(n, per, world)
                n_batches = -(-n // per)
                assert len(by_rank) == world and all(len(rr) == n_batches for rr in by_rank)
                for b in range(n_batches):
                    sizes = [by_rank[r][b][1] - by_rank[r][b][0] for r in range(world)]
                    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == min(per, n - b * per)
                    assert by_rank[0][b][0] == b * per
                    assert all(by_rank[r][b][1] == by_rank[r + 1][b][0] for r in range(world - 1))


def test_shard_partition_properties():
    from metrabs_amd.distributed import shard_internal_batches
    for n in (0, 1, 7, 64, 65):
        for per in (1, 5, 12, 64):
            for world in (1, 2, 8):
                seen = []
                for r in range(world):
                    seen += [i for a, b in shard_internal_batches(n, per, r, world) for i in range(a, b)]
                assert sorted(seen) == list(range(n))  # every box exactly once


def _exact_worker(rank, world, port, n_boxes, per_batch, weak, q):
    """The REAL Pose3dEstimator._predict_in_batches in exact-monolithic mode with the GPU stages
    stubbed: the pyramid is a placeholder and _predict_single_batch is a stand-in crop model that
    makes exactly the collective calls Metrabs.forward makes (one predicate,
    distributed.exact_mode_needs_allreduce).  A mismatch between the two sides -- an empty slice
    all-reducing while the other rank's forward does not -- pairs that all-reduce with the
    other rank's all-gather: a hang or a corrupted gather."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import types
    import numpy as np
    from metrabs_amd import distributed, kernels
    from metrabs_amd.multiperson.multiperson_model import Pose3dEstimator
    distributed.init_from_env(backend='gloo')

    class Crop(torch.nn.Module):
        joint_names = np.array([f'j{i}' for i in range(17)])
        joint_edges = np.array([[i, i + 1] for i in range(16)])
        input_resolution = 256
        exact_monolithic = False
        config = types.SimpleNamespace(weak_perspective=weak)

    est = Pose3dEstimator(Crop(), {'': dict(indices=list(range(17)), names=list(Crop.joint_names),
                                            edges=Crop.joint_edges.tolist())}, None)
    est.shard_across_ranks = 'exact_monolithic'
    calls = []

    def single_batch(pyramid, K, dist12, up, boxes, image_ids, tta, aa, raw=False):
        assert est.crop_model.exact_monolithic
        if distributed.exact_mode_needs_allreduce(est.crop_model):  # == Metrabs.forward
            m = torch.tensor([1.0, 2.0, float(len(boxes))], dtype=torch.float64)
            distributed.allreduce_moments(m)
            calls.append(float(m[2]))
        return boxes[:, :1].reshape(-1, 1, 1, 1).repeat(1, 1, 17, 3).clone()

    est._predict_single_batch = single_batch
    orig = kernels.build_pyramid
    kernels.build_pyramid = lambda images, **kw: None
    try:
        boxes = torch.arange(n_boxes, dtype=torch.float32).reshape(-1, 1).repeat(1, 5)
        tta = dict(gammas=torch.ones(1))
        out = est._predict_in_batches(
            torch.zeros(1, 3, 8, 8, dtype=torch.uint8), torch.eye(3).repeat(n_boxes, 1, 1),
            torch.zeros(n_boxes, 12), torch.zeros(n_boxes, 3), boxes,
            torch.zeros(n_boxes, dtype=torch.int32), per_batch, tta, 1)
    finally:
        kernels.build_pyramid = orig
    q.put((rank, out.numpy(), calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('weak', [False, True])
@pytest.mark.parametrize('n_boxes,per_batch', [(1, 4), (5, 2), (3, 1)])
def test_exact_monolithic_empty_slices_pair_their_collectives(n_boxes, per_batch, weak):
    """Fewer boxes in an internal batch than ranks: one rank's slice is empty.  Full perspective:
    the empty slice joins the batch's moment all-reduce.  Weak perspective (ptu3d.py:36-49 has no
    batch-global scalar): NO all-reduce on either side -- round-2 ADVICE: the empty-slice rank used to
    all-reduce unconditionally and paired with the other rank's all-gather."""
    world = 2
    ctx = mp.get_context('spawn')
    for attempt in range(3):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_exact_worker, args=(r, world, port, n_boxes, per_batch, weak, q))
                 for r in range(world)]
        for p in procs:
            p.start()
        try:
            results = [q.get(timeout=120) for _ in range(world)]
        except Exception:
            results = None
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        if results is not None and all(p.exitcode == 0 for p in procs):
            break
    else:
        pytest.fail('world-size-2 gloo run hung or failed three times (mismatched collectives?)')
    expected = torch.arange(n_boxes, dtype=torch.float32).reshape(-1, 1, 1, 1).repeat(1, 1, 17, 3)
    for rank, out, calls in results:
        assert torch.equal(torch.from_numpy(out), expected), f'rank {rank}: wrong gather'
        if weak:
            assert calls == []
        else:  # every all-reduce this rank's forward joined saw the whole internal batch
            assert all(c in (float(min(per_batch, n_boxes - b)) for b in range(0, n_boxes, per_batch))
                       for c in calls)
