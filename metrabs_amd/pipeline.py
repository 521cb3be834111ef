"""Device-resident per-crop pipeline: geometry -> sampler -> crop model -> mirror un-swap -> back
rotation, i.e. Pose3dEstimator._predict_single_batch (multiperson_model.py:227-259) with every input
already on the GPU and no host synchronisation, so that one internal batch can be captured into a
HIP graph (launch-bound at 64 crops: ~10 kernels of ours + the backbone's).
"""
import torch

from metrabs_amd import kernels

# hipStreamCaptureModeThreadLocal for every capture of this package.  The default ("global") turns a HIP call
# of ANY other thread during the capture into an error -- and once a process group exists, ProcessGroupNCCL's
# watchdog thread polls its events all the time: a capture that overlaps one of its hipEventQuery calls fails
# with "operation not permitted when stream is capturing" INSIDE the watchdog thread, which takes the whole
# process down (seen once in four one-rank RCCL runs of bench.py --graph-gather, round 4).  Thread-local mode
# only polices the capturing thread, which is the one that has to behave.
CAPTURE_ERROR_MODE = 'thread_local'


def predict_single_batch(crop_model, mirror_mapping, should_flip, any_flip, pyramid, intrinsic_matrix,
                         distortion12, camspace_up, boxes, image_ids, rotflipmat, aug_scales,
                         aug_gammas, antialias_factor, crop_dtype=torch.float32,
                         channels_last=False, raw=False):
    """-> poses [n_box, num_aug, J, 3] in the ORIGINAL camera frame.

    All tensor arguments live on the GPU; ``any_flip`` is a host bool (known from num_aug alone)."""
    res = int(crop_model.input_resolution)
    new_k, rot, wp = kernels.crop_geometry(
        boxes, intrinsic_matrix, distortion12, camspace_up, image_ids, rotflipmat, aug_scales,
        aug_gammas, res, antialias_factor)
    crops = kernels.warp_crops(pyramid, wp, res, antialias_factor, out_dtype=crop_dtype,
                               channels_last=channels_last)
    poses_flat = crop_model((crops, new_k.reshape(-1, 3, 3)))
    if raw:
        return poses_flat, rot  # [A*n, J, 3] in crop camera frames + R [A,n,3,3]: K7's inputs
    num_aug = new_k.shape[0]
    poses = poses_flat.reshape(num_aug, -1, poses_flat.shape[-2], 3)
    if any_flip:
        # joints are re-indexed through the mirror mapping BEFORE the back rotation (:249-256)
        swapped = poses[..., mirror_mapping, :]
        poses = torch.where(should_flip.reshape(-1, 1, 1, 1), swapped, poses)
    return (poses @ rot).transpose(0, 1)


class GraphedCropPipeline:
    """One fixed-shape internal batch (n_images frames, n_box boxes, num_aug) captured in a HIP
    graph.  ``run`` copies nothing: callers write into the static input tensors
    (``images``, ``boxes``, ``intrinsics``, ``distortion12``, ``camspace_up``, ``image_ids``,
    ``inv_extrinsics``) and read ``poses3d`` [n_box, J, 3] / ``poses2d`` [n_box, J, 2] (TTA-averaged;
    ``poses`` aliases ``poses3d``)."""

    def __init__(self, estimator, n_images, im_h, im_w, n_box, num_aug=1, antialias_factor=1,
                 use_graph=True, include_pyramid=True):
        dev = estimator._device()
        self.est = estimator
        self.images = torch.zeros(n_images, 3, im_h, im_w, dtype=torch.uint8, device=dev)
        self.boxes = torch.zeros(n_box, 4, device=dev)
        self.boxes[:, 2:] = 100.0
        self.intrinsics = torch.eye(3, device=dev).repeat(n_box, 1, 1)
        self.distortion12 = torch.zeros(n_box, 12, device=dev)
        self.camspace_up = torch.tensor([0.0, -1.0, 0.0], device=dev).repeat(n_box, 1)
        self.image_ids = torch.zeros(n_box, dtype=torch.int32, device=dev)
        self.inv_extrinsics = torch.eye(4, device=dev).repeat(n_box, 1, 1)
        self.average_aug = True
        self.tta = estimator._tta(num_aug, dev)
        self.mirror = self.tta['mirror_i64']
        self.any_flip = bool(self.tta['should_flip_host'].any())
        self.aa = antialias_factor
        self.include_pyramid = include_pyramid
        self.pyramid = None
        self.poses = self.poses3d = self.poses2d = None
        self.graph = None
        self.use_graph = use_graph
        # optional: called with the poses at the end of every step INSIDE the captured region (e.g. the
        # RCCL all-gather of a sharded job, so that a step stays one graph launch); also called by the
        # eager warm-up steps in front of the capture, which is where a communicator initialises
        self.after_step = None

    def _body(self):
        if self.include_pyramid or self.pyramid is None:
            self.pyramid = kernels.build_pyramid(self.images)
        poses_flat, rot = predict_single_batch(
            self.est.crop_model, self.mirror, self.tta['should_flip'], self.any_flip, self.pyramid,
            self.intrinsics, self.distortion12, self.camspace_up, self.boxes, self.image_ids,
            self.tta['rotflipmat'], self.tta['scales'], self.tta['gammas'], self.aa,
            self.est.crop_dtype, self.est.crop_channels_last, raw=True)
        # K7: mirror un-swap, back rotation, 2D projection, world transform, TTA mean in one launch
        self.poses3d, self.poses2d = kernels.postprocess_poses(
            poses_flat, rot, self.tta['should_flip_u8'], self.tta['mirror_i32'], self.intrinsics,
            self.distortion12, self.inv_extrinsics, self.est._joint_transform_on(self.images.device),
            None, self.average_aug)
        if self.after_step is not None:
            self.after_step(self.poses3d)
        return self.poses3d

    def capture(self, warmup=3):
        with torch.inference_mode():
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(warmup):  # lazy inits (MIOpen find, lazy conv, weight packing)
                    self.poses = self._body()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            if self.use_graph:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_ERROR_MODE):
                    self.poses = self._body()
        return self

    def run(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            with torch.inference_mode():
                self.poses = self._body()
        return self.poses
