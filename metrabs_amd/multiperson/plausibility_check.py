"""Drop-in for metrabs_pytorch/multiperson/plausibility_check.py (and its TF twin), backed by the
K8 kernel (csrc/pose_filter.hip).

The reference evaluates the three plausibility tests and the pose NMS as ~40 small tensor ops plus a
Python double loop per image; here `filter_poses` is ONE launch for all images of a call.
The reference reads the mean bone lengths from FLAGS.bone_length_dataset / FLAGS.bone_length_file
(plausibility_check.py:13-16); here they are an argument (`mean_bones`, one length per stick-figure
edge of the model's joint set, mm).

Where the twins differ:
  * variance of the augmentation results: TF reduce_variance = population variance (default here;
    with num_aug=1 every pose is consistent); the PyTorch port's torch.var is unbiased (NaN at
    num_aug=1): `unbiased=True`;
  * order of the survivors: the PyTorch port returns them in ascending index order (default here),
    tf.image.non_max_suppression_overlaps in descending score order, at most 150: `order='score'`.
"""
import torch

from metrabs_amd import kernels


def filter_poses(boxes, poses3d, poses2d, joint_info, mean_bones, unbiased=False, order='index'):
    """_filter_poses (TF multiperson_model.py:441-459).  boxes: list of [n_i,5]; poses3d: list of
    [n_i,A,J,3] camera-space poses of ALL model joints (before the skeleton selection); poses2d:
    list of [n_i,A,J,2].  -> list of int64 index tensors (rows of image i to keep)."""
    counts = [len(b) for b in boxes]
    if sum(counts) == 0:
        return [torch.zeros(0, dtype=torch.int64, device=boxes[0].device if boxes else 'cuda')
                for _ in counts]
    p3, p2, bx = torch.cat(list(poses3d)), torch.cat(list(poses2d)), torch.cat(list(boxes))
    edges = None if mean_bones is None else list(joint_info.stick_figure_edges)
    keep_idx, keep_count, _ = kernels.filter_poses(
        p3, p2, bx, counts, edges, mean_bones, n_joints=joint_info.n_joints, unbiased=unbiased,
        order=order)
    keep_count = keep_count.tolist()  # the only host sync: ragged outputs need the counts
    out, start = [], 0
    for n, k in zip(counts, keep_count):
        out.append((keep_idx[start:start + k] - start).long())
        start += n
    return out
