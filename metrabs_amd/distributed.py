"""Multi-GPU: crops are independent, so the path shards with NO data-path collective; the only
exchange is one all-gather of the resulting poses.

* One process per GPU (torchrun); backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU (tests).
* The shard unit is the reference's INTERNAL BATCH (boxes_per_batch = internal_batch_size // num_aug,
  multiperson_model.py:189-220): reconstruct_ref_fullpersp normalises by an RMS over the whole
  crop_model call (ptu3d.py:71-74), so sharding whole internal batches reproduces the reference's
  numbers with no cross-GPU reduction.  ``exact_monolithic`` is the optional mode in which a caller
  who wants ONE big batch split over ranks all-reduces the three moment scalars between the
  moments and solve kernels.
* The gather payload is KB-sized and latency-bound: a single ``all_gather_into_tensor`` on padded,
  equal-sized shards (xGMI is a full point-to-point mesh; RCCL picks the direct algorithm at this
  size), not a ring of sends.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* (as torchrun sets them).
    Returns (rank, world_size, local_rank).  World size 1 needs no process group."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_internal_batches(n_boxes, boxes_per_batch, rank, world_size):
    """Round-robin assignment of whole internal batches to ranks.

    -> list of (start, stop) box ranges owned by ``rank``, in global order."""
    if boxes_per_batch <= 0:
        boxes_per_batch = max(n_boxes, 1)
    n_batches = (n_boxes + boxes_per_batch - 1) // boxes_per_batch
    return [(i * boxes_per_batch, min((i + 1) * boxes_per_batch, n_boxes))
            for i in range(n_batches) if i % world_size == rank]


def gather_poses(local_poses, ranges_of_rank, n_boxes, boxes_per_batch, world_size, group=None):
    """One all-gather of the per-rank results, then un-shuffling into global box order.

    local_poses: [n_local, ...] results of this rank's internal batches, concatenated in the order
    of ``shard_internal_batches``.  Every rank returns the full [n_boxes, ...] tensor."""
    if world_size == 1:
        return local_poses
    if boxes_per_batch <= 0:
        boxes_per_batch = max(n_boxes, 1)
    n_batches = (n_boxes + boxes_per_batch - 1) // boxes_per_batch
    max_batches = (n_batches + world_size - 1) // world_size
    cap = max_batches * boxes_per_batch  # equal-sized padded shard
    tail = local_poses.shape[1:]
    padded = local_poses.new_zeros((cap,) + tuple(tail))
    padded[:local_poses.shape[0]] = local_poses
    gathered = local_poses.new_empty((world_size * cap,) + tuple(tail))
    dist.all_gather_into_tensor(gathered, padded, group=group)
    gathered = gathered.reshape((world_size, cap) + tuple(tail))
    out = local_poses.new_empty((n_boxes,) + tuple(tail))
    for r in range(world_size):
        offset = 0
        for start, stop in shard_internal_batches(n_boxes, boxes_per_batch, r, world_size):
            out[start:stop] = gathered[r, offset:offset + (stop - start)]
            offset += stop - start
    return out


def allreduce_moments(moments, group=None):
    """exact-monolithic mode: sum the (sum2d, sumrb, count) f64 triple over ranks."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(moments, op=dist.ReduceOp.SUM, group=group)
    return moments
