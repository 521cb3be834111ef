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
