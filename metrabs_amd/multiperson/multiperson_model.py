"""Drop-in for metrabs_pytorch/multiperson/multiperson_model.py (Pose3dEstimator).

Public surface, argument names/defaults, result keys and sentinels follow the reference
(multiperson_model.py:10-13,39-74,384-429; docs/API.md).  The per-crop hot path -- gamma decode +
pyramid, crop geometry, crop sampler, head, reconstruction -- runs in HIP kernels through
metrabs_amd.kernels; everything here is host orchestration with no device synchronisation.

Differences to the reference, all documented in DESIGN.md:
  * the person detector is any callable ``images -> list of [n_i, 5] boxes`` (the reference hard-wires
    an ultralytics YOLOv8, person_detector.py, which is out of scope);
  * ``estimate_poses_batched`` accepts a list of per-image box tensors (the reference's own wrapper
    is broken on current torch, multiperson_model.py:68);
  * images with zero boxes are handled (TF's _predict_empty, metrabs_tf multiperson_model.py:417-439).
"""
import numpy as np
import torch

from metrabs_amd import distributed, graph_cache, kernels, pipeline
from metrabs_amd.joint_info import JointInfo
from metrabs_amd.multiperson import warping

# Dummy value which means that the intrinsic_matrix is unknown (multiperson_model.py:10)
UNKNOWN_INTRINSIC_MATRIX = ((-1, -1, -1), (-1, -1, -1), (-1, -1, -1))
DEFAULT_EXTRINSIC_MATRIX = ((1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1))
DEFAULT_DISTORTION = (0, 0, 0, 0, 0)
DEFAULT_WORLD_UP = (0, -1, 0)


def tta_linspace(start, stop, num, endpoint=True):
    """The reference's linspace for the augmentation tables (ptu.py:78-92): a single endpoint-
    inclusive sample is the MIDPOINT of the range (num_aug = 1: gamma 0.8, angle 0), an
    endpoint-exclusive range stops one step short.  The values have to be torch.linspace's own,
    bit for bit (golden `tta_params`), so it is torch.linspace that produces them."""
    start, stop = torch.as_tensor(start), torch.as_tensor(stop)
    if endpoint and num == 1:
        return ((start + stop) / 2).reshape(1)
    if not endpoint and num > 1:
        stop = stop - (stop - start) / num
    return torch.linspace(start, stop, num)


def tta_parameters(num_aug, rot_aug_degrees=25):
    """Test-time-augmentation table (multiperson_model.py:108-137)."""
    gammas = tta_linspace(np.float32(0.6), np.float32(1.0), num_aug)
    angle_range = np.float32(np.deg2rad(rot_aug_degrees))
    angles = tta_linspace(-angle_range, angle_range, num_aug)
    scales = torch.cat([
        tta_linspace(0.8, 1.0, num_aug // 2, endpoint=False),
        torch.linspace(1.0, 1.1, num_aug - num_aug // 2)], dim=0)
    should_flip = (torch.arange(0, num_aug) - num_aug // 2) % 2 != 0
    flipmat = torch.tensor([[-1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=torch.float32)
    maybe_flipmat = torch.where(should_flip[:, np.newaxis, np.newaxis], flipmat, torch.eye(3))
    rotflipmat = maybe_flipmat @ rotation_about_z(-angles)
    return dict(gammas=gammas, angles=angles, scales=scales, should_flip=should_flip,
                rotflipmat=rotflipmat)


def rotation_about_z(angle):
    """[..., 3, 3] rotation by `angle` about the optical axis, float32 sin / cos as the reference
    evaluates them (ptu3d.py:164-184 with rot_axis='z'; the TTA table is compared bit for bit)."""
    c, s_ = torch.cos(angle), torch.sin(angle)
    o, z = torch.ones_like(angle), torch.zeros_like(angle)
    return torch.stack([c, -s_, z, s_, c, z, z, z, o], dim=-1).reshape(*angle.shape, 3, 3)


def intrinsics_from_fov(fov_degrees, image_hw):
    """[1, 3, 3] pinhole matrix for a field of view over the longer image side, principal point at
    the image centre (ptu3d.py:149-161), in float32 like the reference."""
    h, w = (torch.tensor(float(v), dtype=torch.float32) for v in image_hw)
    half_fov = fov_degrees * torch.tensor(np.pi / 180, dtype=torch.float32) / 2
    f = torch.maximum(h, w) / (torch.tan(half_fov) * 2)
    K = torch.zeros(1, 3, 3)
    K[0, 0, 0] = K[0, 1, 1] = f
    K[0, 0, 2], K[0, 1, 2], K[0, 2, 2] = w / 2, h / 2, 1.0
    return K


def homogeneous(x):
    return torch.cat([x, torch.ones_like(x[..., :1])], dim=-1)


def distort_points(points, coeffs12):
    """warping.distort_points (warping.py:57-62,90-107) as torch ops for the O(n*J) 2D projection of
    the final poses; coeffs12 [n,12], points [n,...,2].  All-zero rows are returned unchanged."""
    d = coeffs12.reshape(coeffs12.shape[0], *([1] * (points.ndim - 2)), 12)
    r2 = torch.sum(torch.square(points), dim=-1, keepdim=True)
    a = ((((d[..., 4:5] * r2 + d[..., 1:2]) * r2 + d[..., 0:1]) * r2 + 1) /
         (((d[..., 7:8] * r2 + d[..., 6:7]) * r2 + d[..., 5:6]) * r2 + 1))
    p2_1 = torch.flip(d[..., 2:4], dims=[-1])
    b = 2 * torch.sum(points * p2_1, dim=-1, keepdim=True)
    c = (d[..., 9:12:2] * r2 + p2_1 + d[..., 8:11:2]) * r2
    distorted = points * (a + b) + c
    has = (d != 0).any(dim=-1, keepdim=True)
    return torch.where(has, distorted, points)


class Pose3dEstimator(torch.nn.Module):
    def __init__(self, crop_model, skeleton_infos, joint_transform_matrix, detector=None):
        super().__init__()
        self.crop_model = crop_model
        self.joint_names = self.crop_model.joint_names
        self.joint_edges = self.crop_model.joint_edges
        self.joint_info = JointInfo(self.joint_names, self.joint_edges)
        self.detector = detector
        self.joint_transform_matrix = (
            None if joint_transform_matrix is None
            else torch.as_tensor(joint_transform_matrix, dtype=torch.float32))
        self.per_skeleton_indices = {
            k: torch.tensor(v['indices'], dtype=torch.int32) for k, v in skeleton_infos.items()}
        self.per_skeleton_joint_names = {k: v['names'] for k, v in skeleton_infos.items()}
        self.per_skeleton_joint_edges = {
            k: torch.tensor(v['edges'], dtype=torch.int32) for k, v in skeleton_infos.items()}
        self.skeleton_joint_indices_table = {k: v['indices'] for k, v in skeleton_infos.items()}
        self._tta_cache = {}
        self.crop_dtype = torch.float32
        self.crop_channels_last = False
        self.shard_across_ranks = False
        self.force_collective = False   # run the final all-gather even in a process group of one rank
        # K7: one HIP launch for everything after the crop model (False = the torch-op sequence)
        self.fused_postprocess = True
        # Mean bone lengths (mm, one per joint_info.stick_figure_edges entry) switch on the
        # plausibility filter + pose NMS for detect_poses*(suppress_implausible_poses=True) -- the TF
        # reference's behaviour (FLAGS.bone_length_file, plausibility_check.py:13-16).  None (the
        # default) = the PyTorch reference's behaviour: the flag is accepted and ignored
        # (multiperson_model.py:158-163 is commented out there).
        self.mean_bone_lengths = None
        self.filter_unbiased_variance = False  # True: torch.var's default, as the PyTorch port
        self.filter_order = 'index'            # 'score': tf.image.non_max_suppression_overlaps order
        # HIP graphs behind the API: an internal batch (geometry -> sampler -> crop model -> K7) of a
        # shape that keeps coming back is captured once and replayed -- at 64 crops the eager path is
        # launch-bound (~10 launches of ours + ~400 of the backbone per batch).  'auto': a shape is
        # captured on its 2nd occurrence; True: on its first; False: never.  A replay gives the eager
        # path's bits (tests/test_gpu_api_graphs.py).  See metrabs_amd/graph_cache.py.
        self.graph_batches = 'auto'
        self.graphs = graph_cache.GraphCache(self)
        # captured graphs read the weights at capture: loading a state dict (into the estimator or the crop
        # model) and moving / casting the module (.to, .cuda, .half: nn.Module._apply) drop them; so do
        # .to / .half on the crop model alone and edits of the head's parameters (graph_cache.BatchGraph.
        # is_current).  In-place edits of single BACKBONE parameters do not: self.graphs.clear() after those.
        drop = lambda *a, **k: self.graphs.clear()
        for m in (self, self.crop_model):
            if hasattr(m, 'register_load_state_dict_post_hook'):
                m.register_load_state_dict_post_hook(drop)
        self._slabs = []            # two pinned staging buffers of the per-box parameters (one H2D copy
        self._slab_turn = 0         # per call; the host may run two calls ahead of the GPU)

    def _apply(self, fn, *args, **kwargs):
        if hasattr(self, 'graphs'):
            self.graphs.clear()   # (the static buffers and captured graphs belong to the old device / dtype)
        return super()._apply(fn, *args, **kwargs)

    # ------------------------------------------------------------------ public API (reference names)

    def detect_poses_batched(
            self, images, intrinsic_matrix=np.array([UNKNOWN_INTRINSIC_MATRIX]),
            distortion_coeffs=np.array([DEFAULT_DISTORTION]),
            extrinsic_matrix=np.array([DEFAULT_EXTRINSIC_MATRIX]),
            world_up_vector=DEFAULT_WORLD_UP, default_fov_degrees=55, internal_batch_size=64,
            antialias_factor=1, num_aug=5, average_aug=True, skeleton='', detector_threshold=0.3,
            detector_nms_iou_threshold=0.7, max_detections=None, detector_flip_aug=False,
            suppress_implausible_poses=True):
        if self.detector is None:
            raise RuntimeError('no person detector attached: pass detector=callable to '
                               'Pose3dEstimator or use estimate_poses*(images, boxes)')
        boxes = self.detector(
            images=images, threshold=detector_threshold,
            nms_iou_threshold=detector_nms_iou_threshold, max_detections=max_detections)
        return self._estimate_poses_batched(
            images, boxes, intrinsic_matrix, distortion_coeffs, extrinsic_matrix, world_up_vector,
            default_fov_degrees, internal_batch_size, antialias_factor, num_aug, average_aug,
            skeleton, suppress_implausible_poses)

    def estimate_poses_batched(
            self, images, boxes, intrinsic_matrix=(UNKNOWN_INTRINSIC_MATRIX,),
            distortion_coeffs=(DEFAULT_DISTORTION,),
            extrinsic_matrix=(DEFAULT_EXTRINSIC_MATRIX,), world_up_vector=DEFAULT_WORLD_UP,
            default_fov_degrees=55, internal_batch_size=64, antialias_factor=1, num_aug=5,
            average_aug=True, skeleton=''):
        boxes = [torch.as_tensor(b, dtype=torch.float32) for b in boxes]
        boxes = [b.reshape(-1, 4) if b.numel() == 0 else b for b in boxes]  # (an empty (0,) list of boxes)
        boxes = [torch.cat([b[..., :4], torch.ones_like(b[..., :1])], dim=-1) for b in boxes]
        pred = self._estimate_poses_batched(
            images, boxes, intrinsic_matrix, distortion_coeffs, extrinsic_matrix, world_up_vector,
            default_fov_degrees, internal_batch_size, antialias_factor, num_aug, average_aug,
            skeleton, suppress_implausible_poses=False)
        del pred['boxes']
        return pred

    def detect_poses(
            self, image, intrinsic_matrix=UNKNOWN_INTRINSIC_MATRIX,
            distortion_coeffs=DEFAULT_DISTORTION, extrinsic_matrix=DEFAULT_EXTRINSIC_MATRIX,
            world_up_vector=DEFAULT_WORLD_UP, default_fov_degrees=55, internal_batch_size=64,
            antialias_factor=1, num_aug=5, average_aug=True, skeleton='', detector_threshold=0.3,
            detector_nms_iou_threshold=0.7, max_detections=-1, detector_flip_aug=False,
            suppress_implausible_poses=True):
        result = self.detect_poses_batched(
            image[np.newaxis], _one(intrinsic_matrix), _one(distortion_coeffs),
            _one(extrinsic_matrix), world_up_vector, default_fov_degrees, internal_batch_size,
            antialias_factor, num_aug, average_aug, skeleton, detector_threshold,
            detector_nms_iou_threshold, max_detections, detector_flip_aug,
            suppress_implausible_poses)
        return {k: v[0] for k, v in result.items()}

    def estimate_poses(
            self, image, boxes, intrinsic_matrix=UNKNOWN_INTRINSIC_MATRIX,
            distortion_coeffs=DEFAULT_DISTORTION, extrinsic_matrix=DEFAULT_EXTRINSIC_MATRIX,
            world_up_vector=DEFAULT_WORLD_UP, default_fov_degrees=55, internal_batch_size=64,
            antialias_factor=1, num_aug=5, average_aug=True, skeleton=''):
        result = self.estimate_poses_batched(
            image[np.newaxis], [torch.as_tensor(boxes, dtype=torch.float32)],
            _one(intrinsic_matrix), _one(distortion_coeffs), _one(extrinsic_matrix),
            world_up_vector, default_fov_degrees, internal_batch_size, antialias_factor, num_aug,
            average_aug, skeleton)
        return {k: v[0] for k, v in result.items()}

    # ------------------------------------------------------------------ implementation

    def _device(self):
        try:
            return next(self.crop_model.parameters()).device
        except StopIteration:
            return torch.device('cuda')

    def _skeleton_tensor(self, skeleton, dev):
        key = ('skel', skeleton, str(dev))
        if key not in self._tta_cache:
            self._tta_cache[key] = torch.as_tensor(
                self.skeleton_joint_indices_table[skeleton], dtype=torch.int32, device=dev)
        return self._tta_cache[key]

    def _joint_transform_on(self, dev):
        if self.joint_transform_matrix is None:
            return None
        key = ('jtm', str(dev))
        if key not in self._tta_cache:
            self._tta_cache[key] = self.joint_transform_matrix.to(dev, torch.float32).contiguous()
        return self._tta_cache[key]

    def _tta(self, num_aug, dev):
        key = (num_aug, str(dev))
        if key not in self._tta_cache:
            t = tta_parameters(num_aug)
            self._tta_cache[key] = {k: v.to(dev) for k, v in t.items()}
            self._tta_cache[key]['should_flip_host'] = t['should_flip']
            # kernel-ready copies (no per-call casts / host->device copies)
            self._tta_cache[key]['should_flip_u8'] = t['should_flip'].to(dev, torch.uint8)
            self._tta_cache[key]['mirror_i32'] = torch.as_tensor(
                self.joint_info.mirror_mapping, dtype=torch.int32, device=dev)
            self._tta_cache[key]['mirror_i64'] = self._tta_cache[key]['mirror_i32'].long()
        return self._tta_cache[key]

    @torch.no_grad()
    def _estimate_poses_batched(
            self, images, boxes, intrinsic_matrix, distortion_coeffs, extrinsic_matrix,
            world_up_vector, default_fov_degrees, internal_batch_size, antialias_factor, num_aug,
            average_aug, skeleton, suppress_implausible_poses):
        """multiperson_model.py:76-182."""
        dev = self._device()
        if dev.type != 'cuda':
            raise RuntimeError('metrabs_amd.Pose3dEstimator needs the crop model on a GPU')
        antialias_factor = int(antialias_factor)
        if antialias_factor < 1 or antialias_factor == 3 or antialias_factor > 19:
            # the reference shrinks only for 2, 4 and > 4 (multiperson_model.py:307-315): at 3 its
            # reshape to [num_aug, n, 3, res, res] fails; > 19 exceeds the shrink filter's 40 taps
            raise ValueError(f'antialias_factor must be 1, 2, 4 or 5..19 (got {antialias_factor})')
        # frames stay where the caller has them (host or device): they are copied straight into the
        # buffer the sampler reads (_predict_in_batches)
        images = torch.as_tensor(images)
        n_images = len(images)
        intrinsic_matrix = _as_f32(intrinsic_matrix)  # (camera set-up happens on the host)
        distortion_coeffs = _as_f32(distortion_coeffs)
        extrinsic_matrix = _as_f32(extrinsic_matrix)
        world_up_vector = _as_f32(world_up_vector)

        # camera set-up on the host (tiny), then ONE transfer (multiperson_model.py:79-105)
        if len(intrinsic_matrix) == 1:
            if torch.all(intrinsic_matrix == -1):
                intrinsic_matrix = intrinsics_from_fov(default_fov_degrees, images.shape[2:4])
            intrinsic_matrix = intrinsic_matrix.expand(n_images, 3, 3)
        if len(distortion_coeffs) == 1:
            distortion_coeffs = distortion_coeffs.expand(n_images, -1)
        if len(extrinsic_matrix) == 1:
            extrinsic_matrix = extrinsic_matrix.expand(n_images, 4, 4)
        counts = [len(b) for b in boxes]
        n_total = sum(counts)
        camspace_up = torch.einsum('c,bCc->bC', world_up_vector, extrinsic_matrix[..., :3, :3])
        inv_extrinsics = torch.linalg.inv(extrinsic_matrix)
        image_id_host = np.repeat(np.arange(n_images), counts)
        boxes_out = boxes
        boxes_host = (torch.cat([torch.as_tensor(b, dtype=torch.float32).reshape(-1, 5)
                                 for b in boxes if len(b)], dim=0).cpu().numpy() if n_total
                      else np.zeros((0, 5), np.float32))
        per_box = lambda x: np.ascontiguousarray(x, dtype=np.float32).reshape(
            n_images, int(np.prod(x.shape[1:])))[image_id_host]   # (-1 cannot be inferred for 0 frames)
        (boxes_flat, intrinsic_matrix_b, distortion_b, camspace_up_b, image_id_per_box,
         inv_extrinsics_b) = self._upload_per_box(dev, [
            (boxes_host, (5,)), (per_box(intrinsic_matrix.numpy()), (3, 3)),
            (per_box(warping.pad_axis_to_size(distortion_coeffs, 12).numpy()), (12,)),
            (per_box(camspace_up.numpy()), (3,)), (image_id_host.astype(np.int32), ()),
            (per_box(inv_extrinsics.numpy()), (4, 4))])

        tta = self._tta(num_aug, dev)
        n_joints = self.joint_info.n_joints
        idx = self.skeleton_joint_indices_table[skeleton]
        suppress = bool(suppress_implausible_poses) and self.mean_bone_lengths is not None
        if self.fused_postprocess and not suppress:
            n_out = len(idx)
            if sum(counts) == 0:
                shape = (0, n_out) if average_aug else (0, num_aug, n_out)
                poses3d_flat = torch.zeros(*shape, 3, device=dev)
                poses2d_flat = torch.zeros(*shape, 2, device=dev)
            else:
                post = dict(inv_extrinsics=inv_extrinsics_b, average_aug=average_aug,
                            skeleton=self._skeleton_tensor(skeleton, dev),
                            joint_transform=self._joint_transform_on(dev))
                packed = self._predict_in_batches(
                    images, intrinsic_matrix_b, distortion_b, camspace_up_b, boxes_flat,
                    image_id_per_box, internal_batch_size, tta, antialias_factor, post=post,
                    image_id_host=image_id_host)
                poses3d_flat, poses2d_flat = packed[..., :3], packed[..., 3:]
        else:
            if sum(counts) == 0:
                poses3d_flat = torch.zeros(0, num_aug, n_joints, 3, device=dev)
            else:
                poses3d_flat = self._predict_in_batches(
                    images, intrinsic_matrix_b, distortion_b, camspace_up_b, boxes_flat,
                    image_id_per_box, internal_batch_size, tta, antialias_factor,
                    image_id_host=image_id_host)
            # post-processing as torch ops (multiperson_model.py:143-178)
            if self.joint_transform_matrix is not None:
                poses3d_flat = torch.einsum(
                    'bank,nN->baNk', poses3d_flat, self.joint_transform_matrix.to(dev))
            poses2d_flat_normalized = homogeneous(
                distort_points(poses3d_flat[..., :2] / poses3d_flat[..., 2:3], distortion_b))
            poses2d_flat = torch.einsum(
                'bank,bjk->banj', poses2d_flat_normalized, intrinsic_matrix_b[:, :2, :])
            if suppress and sum(counts):
                # TF multiperson_model.py:404-409,441-459: filter on the camera-space poses of all
                # joints, before the world transform and the skeleton selection (K8, one launch)
                from metrabs_amd.multiperson import plausibility_check
                boxes_dev = list(torch.split(boxes_flat, counts))
                keep = plausibility_check.filter_poses(
                    boxes_dev, list(torch.split(poses3d_flat, counts)),
                    list(torch.split(poses2d_flat, counts)), self.joint_info, self.mean_bone_lengths,
                    unbiased=self.filter_unbiased_variance, order=self.filter_order)
                offsets = np.cumsum([0] + counts[:-1])
                sel = torch.cat([k + int(o) for k, o in zip(keep, offsets)])
                boxes_out = [b[k] for b, k in zip(boxes_dev, keep)]
                poses3d_flat, poses2d_flat = poses3d_flat[sel], poses2d_flat[sel]
                inv_extrinsics_b = inv_extrinsics_b[sel]
                counts = [len(k) for k in keep]
            poses3d_flat = torch.einsum(
                'bank,bjk->banj', homogeneous(poses3d_flat), inv_extrinsics_b[:, :3, :])
            poses3d_flat = poses3d_flat[..., idx, :]
            poses2d_flat = poses2d_flat[..., idx, :]
            if average_aug:
                poses3d_flat = torch.mean(poses3d_flat, dim=-3)
                poses2d_flat = torch.mean(poses2d_flat, dim=-3)
        poses3d = list(torch.split(poses3d_flat, counts))
        poses2d = list(torch.split(poses2d_flat, counts))
        return dict(boxes=boxes_out, poses3d=poses3d, poses2d=poses2d)

    def _upload_per_box(self, dev, fields):
        """[(host array [n, ...] f32 or [n] i32, row shape), ...] -> the same as device tensors, through
        ONE pinned staging buffer and ONE host-to-device copy (field after field: every field, and every
        box range of it, is a contiguous view of the device copy)."""
        n = len(fields[0][0])
        sizes = [int(np.prod(shape, dtype=np.int64)) * n for _, shape in fields]
        total = max(sum(sizes), 1)
        if not self._slabs:
            self._slabs = [[None, None], [None, None]]
        self._slab_turn ^= 1
        slot = self._slabs[self._slab_turn]
        if slot[0] is None or slot[0].numel() < total:
            slot[0] = torch.empty(max(total, 4096), dtype=torch.float32).pin_memory()
            slot[1] = torch.cuda.Event()
        else:
            slot[1].synchronize()  # (the copy of the call before last has read this staging buffer)
        stage = slot[0].numpy()
        off = 0
        for (arr, _), size in zip(fields, sizes):
            dst = stage[off:off + size]
            if arr.dtype == np.int32:
                dst.view(np.int32)[:] = arr.reshape(-1)
            else:
                dst[:] = arr.reshape(-1)
            off += size
        slab = torch.empty(total, dtype=torch.float32, device=dev)
        slab.copy_(slot[0][:total], non_blocking=True)
        slot[1].record(torch.cuda.current_stream(dev))
        out, off = [], 0
        for (arr, shape), size in zip(fields, sizes):
            t = slab[off:off + size]
            if arr.dtype == np.int32:
                t = t.view(torch.int32)
            out.append(t.reshape(n, *shape))
            off += size
        return out

    def _predict_in_batches(self, images, intrinsic_matrix, distortion12, camspace_up, boxes_flat,
                            image_id_per_box, internal_batch_size, tta, antialias_factor, post=None,
                            image_id_host=None):
        """multiperson_model.py:184-225.  The whole-image gamma decode (:196) is fused with the
        pyramid build: one launch for all images of the call.  With ``shard_across_ranks`` the
        internal batches are dealt round-robin to the ranks of the default process group and the
        results are all-gathered once at the end (metrabs_amd/distributed.py);
        ``shard_across_ranks = 'exact_monolithic'`` instead cuts EVERY internal batch into one slice
        per rank and all-reduces the three reconstruction moments, i.e. the numbers of the
        un-sharded call.  Either way a rank builds the pyramid of the frames its own boxes reference.

        post=None  -> poses [n, A, J, 3] in the original camera frame (torch-op post-processing);
        post=dict  -> K7 runs per internal batch; returns [n, (A,) S, 5] = poses3d | poses2d."""
        num_aug = len(tta['gammas'])
        boxes_per_batch = internal_batch_size // num_aug
        n_total = len(boxes_flat)
        if boxes_per_batch == 0:
            boxes_per_batch = n_total
        rank, world = 0, 1
        if self.shard_across_ranks and torch.distributed.is_initialized():
            rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
        exact = self.shard_across_ranks == 'exact_monolithic' and world > 1
        # ONE predicate on both sides of the moment all-reduce (Metrabs.forward, models/metrabs.py):
        # the weak-perspective reference point (ptu3d.py:36-49) has no batch-global scalar, so a
        # weak-perspective model in exact mode is plain slicing with no collective at all
        exact_allreduce = exact and distributed.exact_mode_needs_allreduce(self.crop_model)
        if exact:
            ranges_by_rank = distributed.split_internal_batches(n_total, boxes_per_batch, world)
        else:
            ranges_by_rank = [distributed.shard_internal_batches(n_total, boxes_per_batch, r, world)
                              for r in range(world)]
        if image_id_host is not None and images.dtype == torch.uint8 and images.numel() >= kernels.MAX_U8_FRAME_BYTES:
            # the sampler's limit, checked for EVERY rank's ranges before any kernel or collective runs
            # (and before the frames are cut down to this rank's), so that all ranks
            # raise the same error together (one rank raising inside the loop would leave the others
            # waiting in the moment all-reduce / the final gather)
            frame_bytes = int(np.prod(images.shape[1:]))
            for rr in ranges_by_rank:
                for a_, b_ in rr:
                    n_ref = len(set(image_id_host[a_:b_].tolist()))
                    if n_ref * frame_bytes >= kernels.MAX_U8_FRAME_BYTES:
                        raise ValueError(
                            f'one internal batch references {n_ref} frames = {n_ref * frame_bytes} bytes; '
                            f'the sampler takes < {kernels.MAX_U8_FRAME_BYTES} bytes of uint8 frames per '
                            f'call: lower internal_batch_size')
        ranges = ranges_by_rank[rank]
        # the pyramid of the frames THIS rank's boxes reference (all of them on one rank)
        if world > 1 and image_id_host is not None:
            needed = sorted({int(i) for a, b in ranges for i in image_id_host[a:b]})
            if len(needed) < len(images):
                remap = torch.full((len(images),), -1, dtype=image_id_per_box.dtype)
                remap[needed] = torch.arange(len(needed), dtype=image_id_per_box.dtype)
                image_id_per_box = remap.to(image_id_per_box.device)[image_id_per_box.long()]
                images = images[torch.tensor(needed, device=images.device)] if needed else images[:0]
        # mtr_warp_crops_u8 addresses the uint8 frames of a call through one 32-bit-offset buffer
        # descriptor (< 2 GiB: 86 4K frames).  Beyond that every internal batch gets the pyramid of
        # the frames ITS boxes reference (at most boxes_per_batch of them).
        per_batch_pyramids = images.numel() >= kernels.MAX_U8_FRAME_BYTES
        dev = boxes_flat.device
        # frames of another dtype than uint8 (legal for the reference: `images.float() / 255`, :196) take the
        # materialised f32 level 0 and stay outside the static uint8 frame sets / graphs
        u8_frames = images.dtype == torch.uint8
        # HIP graphs (graph_cache.py): the internal batches of this call whose shape has a captured
        # graph, or is due for one, replay it; the others run the same launches eagerly
        plan = None
        if (post is not None and not exact and not per_batch_pyramids and len(images) and dev.type == 'cuda'
                and u8_frames and self.graph_batches and type(self)._predict_single_batch is Pose3dEstimator._predict_single_batch
                and '_predict_single_batch' not in self.__dict__):
            plan = self.graphs.plan_call(images, ranges, tta, antialias_factor, post)
        staging = None
        if (plan is None and dev.type == 'cuda' and not images.is_cuda and images.is_pinned() and len(images)
                and not per_batch_pyramids and u8_frames):
            # pinned host frames: the PCIe copy runs on a copy stream under the previous call's compute
            # (None: no frame set free for this frame size right now -- the plain blocking upload below)
            staging = self.graphs.frame_set(len(images), images.shape[2], images.shape[3], dev, optional=True,
                                            hwc=kernels.frames_are_interleaved(images))
        if plan is not None:
            pyramid = plan.frames.load(images)   # static frame + pyramid buffers the graphs read
        elif staging is not None:
            pyramid = staging.load(images)
        else:
            images = images.to(dev)
            pyramid = kernels.pyramid_of_frames(images) if len(images) and not per_batch_pyramids else None
        if exact:
            if not hasattr(self.crop_model, 'exact_monolithic'):
                raise RuntimeError("shard_across_ranks='exact_monolithic' needs metrabs_amd's Metrabs "
                                   "crop model (the moments are all-reduced inside its forward)")
            self.crop_model.exact_monolithic = True
        out = []
        try:
            for i_range, (start, stop) in enumerate(ranges):
                if start == stop:  # (exact mode) an empty slice still joins the batch's all-reduce
                    if exact_allreduce:
                        distributed.allreduce_moments(torch.zeros(3, dtype=torch.float64,
                                                                  device=boxes_flat.device))
                    continue
                s = slice(start, stop)
                batch_pyramid, batch_ids = pyramid, image_id_per_box[s]
                if per_batch_pyramids:
                    batch_pyramid, batch_ids = self._pyramid_of_referenced_frames(images, batch_ids)
                if post is not None:
                    batch_args = (intrinsic_matrix[s], distortion12[s], camspace_up[s], boxes_flat[s], batch_ids,
                                  post['inv_extrinsics'][s])
                    graph = plan.graph_for(i_range, batch_args) if plan is not None else None
                    if graph is not None:
                        res = graph.replay(batch_args)
                    else:
                        res = self._batch_with_postprocess(batch_pyramid, *batch_args, tta, antialias_factor, post)
                else:
                    res = self._predict_single_batch(
                        batch_pyramid, intrinsic_matrix[s], distortion12[s], camspace_up[s], boxes_flat[s],
                        batch_ids, tta, antialias_factor, raw=False)
                out.append(res)
        finally:
            if exact:
                self.crop_model.exact_monolithic = False
        if out:
            local = torch.cat(out, dim=0)
        elif post is not None:
            n_out = post['skeleton'].numel()
            shape = (0, n_out, 5) if post['average_aug'] else (0, num_aug, n_out, 5)
            local = torch.zeros(*shape, device=boxes_flat.device)
        else:
            local = torch.zeros(0, num_aug, self.joint_info.n_joints, 3, device=boxes_flat.device)
        return distributed.gather_ranges(local, ranges_by_rank, n_total, always=self.force_collective)

    def _batch_with_postprocess(self, pyramid, intrinsic_matrix, distortion12, camspace_up, boxes, image_ids,
                                inv_extrinsics, tta, antialias_factor, post):
        """One internal batch up to and including K7 -> [n, (A,) S, 5] = poses3d | poses2d.  THE body of
        the fused path: run eagerly here and captured, call for call, by graph_cache.BatchGraph."""
        poses_flat, rot = self._predict_single_batch(
            pyramid, intrinsic_matrix, distortion12, camspace_up, boxes, image_ids, tta, antialias_factor,
            raw=True)
        p3, p2 = kernels.postprocess_poses(
            poses_flat, rot, tta['should_flip_u8'], tta['mirror_i32'], intrinsic_matrix, distortion12,
            inv_extrinsics, post['joint_transform'], post['skeleton'], post['average_aug'])
        return torch.cat([p3, p2], dim=-1)

    @staticmethod
    def _pyramid_of_referenced_frames(images, image_ids):
        """-> (pyramid of the frames `image_ids` reference, the ids renumbered into it)."""
        needed, local_ids = torch.unique(image_ids.long(), sorted=True, return_inverse=True)
        frames = images[needed.to(images.device)].to(image_ids.device)
        if frames.dtype == torch.uint8 and frames.numel() >= kernels.MAX_U8_FRAME_BYTES:
            raise ValueError(
                f'one internal batch references {len(needed)} frames = {frames.numel()} bytes; the '
                f'sampler takes < {kernels.MAX_U8_FRAME_BYTES} bytes of uint8 frames per call: lower '
                f'internal_batch_size')
        return kernels.pyramid_of_frames(frames), local_ids.to(image_ids.dtype)

    def _get_crops(self, pyramid, intrinsic_matrix, distortion12, camspace_up, boxes, image_ids, tta,
                   antialias_factor):
        """multiperson_model.py:264-320: two launches (geometry, sampler) for the internal batch."""
        res = int(self.crop_model.input_resolution)
        new_k, rot, wp = kernels.crop_geometry(
            boxes, intrinsic_matrix, distortion12, camspace_up, image_ids, tta['rotflipmat'],
            tta['scales'], tta['gammas'], res, antialias_factor)
        crops = kernels.warp_crops(pyramid, wp, res, antialias_factor, out_dtype=self.crop_dtype,
                                   channels_last=self.crop_channels_last)
        return crops, new_k, rot

    def _predict_single_batch(self, pyramid, intrinsic_matrix, distortion12, camspace_up, boxes,
                              image_ids, tta, antialias_factor, raw=False):
        """multiperson_model.py:227-259 (raw=True stops after the crop model: K7 does the rest)."""
        return pipeline.predict_single_batch(
            self.crop_model, tta['mirror_i64'], tta['should_flip'], bool(tta['should_flip_host'].any()),
            pyramid, intrinsic_matrix, distortion12, camspace_up, boxes, image_ids,
            tta['rotflipmat'], tta['scales'], tta['gammas'], antialias_factor, self.crop_dtype,
            self.crop_channels_last, raw=raw)


def _as_f32(x):
    if torch.is_tensor(x):
        return x.detach().to('cpu', torch.float32)
    return torch.as_tensor(np.asarray(x), dtype=torch.float32)


def _one(x):
    return _as_f32(x)[np.newaxis]
