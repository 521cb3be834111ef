"""CPU, world sizes 2 AND 8 over gloo: the N>1 path -- whole internal batches dealt round-robin to ranks,
no data-path collective, one all-gather of the poses at the end, optional all-reduce of the three
reconstruction moments (exact-monolithic mode).  World 8 is what the driver's scaling run launches
(BASELINE.json configs[2]: 256 crops, internal batch 32, one batch per rank); every case asserts that each
rank's gather equals the single-rank result and that all ranks issued the same collectives."""
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


def _worker(rank, world, port, n_boxes, per_batch, exact, q):
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


def _count_collectives(counter):
    """Wraps the two collectives the path may issue so that a worker can report how many it joined."""
    orig_gather, orig_reduce = dist.all_gather_into_tensor, dist.all_reduce

    def gather(*a, **k):
        counter['all_gather'] += 1
        return orig_gather(*a, **k)

    def reduce(*a, **k):
        counter['all_reduce'] += 1
        return orig_reduce(*a, **k)

    dist.all_gather_into_tensor, dist.all_reduce = gather, reduce


def _run_ranks(target, world, args, what):
    ctx = mp.get_context('spawn')
    for attempt in range(3):  # (the probed port can be taken between the probe and the rendezvous)
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            results = [q.get(timeout=90) for _ in range(world)]
        except Exception:
            results = None
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        if results is not None and all(p.exitcode == 0 for p in procs):
            return sorted(results, key=lambda r: r[0])
    pytest.fail(f'world-size-{world} gloo run {what} three times')


@pytest.mark.parametrize('world,n_boxes,per_batch,exact', [
    (2, 23, 4, False), (2, 8, 4, False), (2, 3, 12, False), (2, 5, 1, False), (2, 23, 4, True), (2, 5, 1, True),
    (2, 7, 64, True),
    (8, 256, 32, False),   # BASELINE configs[2]: 256 crops, internal batch 32 -> one batch per rank
    (8, 100, 7, False),    # ragged: 15 batches over 8 ranks, a 2-box tail
    (8, 3, 12, False),     # one batch: seven ranks own nothing and still join the gather
    (8, 37, 5, True),      # exact-monolithic: 5-box batches over 8 ranks (three empty slices each), a 2-box tail
    (8, 5, 64, True)])     # fewer boxes than ranks in the only batch
def test_round_robin_shards_and_single_gather(world, n_boxes, per_batch, exact):
    results = _run_ranks(_worker, world, (n_boxes, per_batch, exact), 'failed')
    expected = torch.arange(n_boxes, dtype=torch.float32).reshape(-1, 1, 1, 1).repeat(1, 2, 17, 3) \
        + 0.001 * (torch.arange(n_boxes) // per_batch).float().reshape(-1, 1, 1, 1)
    for rank, full, moments in results:
        full, moments = torch.from_numpy(full), torch.from_numpy(moments)
        assert full.shape == (n_boxes, 2, 17, 3)
        assert torch.equal(full, expected), f'rank {rank} gathered a wrong / mis-ordered result'
        tri = world * (world + 1) / 2
        assert moments[:2].tolist() == [tri, 10.0 * tri] and moments[2] == n_boxes


def test_unshuffle_index_is_the_slice_copy_loop():
    """gather_ranges un-shuffles with ONE index_select by a host-built index (round 4 copied a slice per
    range per rank): the index equals what the loop did, for round-robin and exact-monolithic ranges, a
    thousand internal batches included."""
    import numpy as np
    from metrabs_amd.distributed import shard_internal_batches, split_internal_batches, unshuffle_index
    for n, per, world in ((256, 32, 8), (100, 7, 8), (3, 12, 8), (1000, 1, 8), (37, 5, 3), (0, 4, 2)):
        for by_rank in ([shard_internal_batches(n, per, r, world) for r in range(world)],
                        split_internal_batches(n, per, world)):
            cap = max(sum(b - a for a, b in rr) for rr in by_rank) if n else 0
            want = np.full(n, -1, np.int64)
            for r, rr in enumerate(by_rank):
                off = 0
                for a, b in rr:
                    want[a:b] = r * cap + off + np.arange(b - a)
                    off += b - a
            assert np.array_equal(unshuffle_index(by_rank, cap, n), want)
    with pytest.raises(ValueError):
        unshuffle_index([[(0, 2)], [(3, 4)]], 2, 4)   # box 2 belongs to nobody


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


def _estimator_worker(rank, world, port, n_boxes, per_batch, weak, mode, q):
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
    counter = dict(all_gather=0, all_reduce=0)
    _count_collectives(counter)

    class Crop(torch.nn.Module):
        joint_names = np.array([f'j{i}' for i in range(17)])
        joint_edges = np.array([[i, i + 1] for i in range(16)])
        input_resolution = 256
        exact_monolithic = False
        config = types.SimpleNamespace(weak_perspective=weak)

    est = Pose3dEstimator(Crop(), {'': dict(indices=list(range(17)), names=list(Crop.joint_names),
                                            edges=Crop.joint_edges.tolist())}, None)
    est.shard_across_ranks = mode
    calls = []

    def single_batch(pyramid, K, dist12, up, boxes, image_ids, tta, aa, raw=False):
        assert est.crop_model.exact_monolithic == (mode == 'exact_monolithic')
        if mode == 'exact_monolithic' and distributed.exact_mode_needs_allreduce(est.crop_model):  # == Metrabs.forward
            m = torch.tensor([1.0, 2.0, float(len(boxes))], dtype=torch.float64)
            distributed.allreduce_moments(m)
            calls.append(float(m[2]))
        # (the batch size rides along: round-robin ranks must have seen WHOLE internal batches)
        return (boxes[:, :1] + 0.001 * len(boxes)).reshape(-1, 1, 1, 1).repeat(1, 1, 17, 3).clone()

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
    q.put((rank, out.numpy(), calls, counter))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
@pytest.mark.parametrize('weak', [False, True])
@pytest.mark.parametrize('n_boxes,per_batch', [(1, 4), (5, 2), (3, 1)])
def test_exact_monolithic_empty_slices_pair_their_collectives(n_boxes, per_batch, weak, world):
    """Fewer boxes in an internal batch than ranks: some ranks' slices are empty.  Full perspective:
    the empty slice joins the batch's moment all-reduce.  Weak perspective (ptu3d.py:36-49 has no
    batch-global scalar): NO all-reduce on either side -- round-2 ADVICE: the empty-slice rank used to
    all-reduce unconditionally and paired with the other rank's all-gather.  World 8: at most 2 of the 8
    ranks hold a box of a batch."""
    results = _run_ranks(_estimator_worker, world, (n_boxes, per_batch, weak, 'exact_monolithic'),
                         'hung or failed (mismatched collectives?)')
    n_batches = -(-n_boxes // per_batch)
    for rank, out, calls, counter in results:
        # every rank: the un-sharded result (pose of box i = i + 0.001 * its slice's size on its rank)
        got = torch.from_numpy(out)
        assert got.shape == (n_boxes, 1, 17, 3) and torch.equal(got.floor(), torch.arange(
            n_boxes, dtype=torch.float32).reshape(-1, 1, 1, 1).repeat(1, 1, 17, 3)), f'rank {rank}: wrong gather'
        # every rank issued the same collectives: one all-reduce per internal batch (none for weak
        # perspective), one all-gather
        assert counter == dict(all_gather=1, all_reduce=0 if weak else n_batches), (rank, counter)
        if weak:
            assert calls == []
        else:  # every all-reduce this rank's forward joined saw the whole internal batch
            assert all(c in (float(min(per_batch, n_boxes - b)) for b in range(0, n_boxes, per_batch))
                       for c in calls)


@pytest.mark.parametrize('world,n_boxes,per_batch', [(8, 256, 32), (8, 50, 4), (8, 3, 2), (2, 50, 4)])
def test_round_robin_estimator_at_world_8(world, n_boxes, per_batch):
    """Pose3dEstimator._predict_in_batches with shard_across_ranks=True (the default sharded mode, SURVEY.md
    section 8e / multiperson_model.py:189-220) at the driver's world size: every rank runs WHOLE internal
    batches (the stand-in crop model writes its batch size into the pose), issues exactly one collective
    (the final all-gather) and returns the single-rank result."""
    results = _run_ranks(_estimator_worker, world, (n_boxes, per_batch, False, True), 'hung or failed')
    sizes = torch.tensor([min(per_batch, n_boxes - (i // per_batch) * per_batch) for i in range(n_boxes)])
    expected = (torch.arange(n_boxes, dtype=torch.float32) + 0.001 * sizes).reshape(-1, 1, 1, 1).repeat(1, 1, 17, 3)
    for rank, out, calls, counter in results:
        assert torch.equal(torch.from_numpy(out), expected), f'rank {rank}: wrong gather'
        assert counter == dict(all_gather=1, all_reduce=0) and calls == []
