"""Drop-in for metrabs_pytorch/models/metrabs.py: Metrabs (crop model) and MetrabsHeads.

Same constructor roles and forward signatures as the reference; the head runs in hand-written HIP:
the fused 1x1-projection GEMM + soft-argmax decode (csrc/head_fused.hip) or, with
``fused=False``, a library GEMM for the 1x1 conv followed by the HIP decode kernel
(csrc/decode.hip).  There is no CPU path."""
import os
import threading

import numpy as np
import torch

from metrabs_amd import distributed, kernels
from metrabs_amd.config import MetrabsConfig


class MetrabsHeads(torch.nn.Module):
    """models/metrabs.py:67-85.  ``conv_final`` keeps the reference's parameter names and layout
    (weight [J*(1+D), C, 1, 1], bias [J*(1+D)]; state_dict key heatmap_heads.conv_final.*,
    convert_model_from_tf.py:175-177,194) so reference checkpoints load unchanged."""

    def __init__(self, n_points, config=None, in_channels=None, fused=True):
        super().__init__()
        self.config = MetrabsConfig.from_any(config) if config is not None else MetrabsConfig()
        self.n_points = n_points
        self.n_outs = [n_points, self.config.depth * n_points]
        if in_channels is None:
            self.conv_final = torch.nn.LazyConv2d(out_channels=sum(self.n_outs), kernel_size=1)
        else:
            self.conv_final = torch.nn.Conv2d(in_channels, sum(self.n_outs), kernel_size=1)
        # True: hand-written GEMM + decode in one kernel.  False: library 1x1 conv + the HIP decode
        # kernel on the materialised logits.  'auto' (Metrabs' default): a STATIC rule on (dtype,
        # layout, C, H, W, J, D) -- never on the batch size, never on a clock (kernels.head_auto_choice):
        # the same deployment runs the same kernel, and therefore gives the same bits, in every
        # process, on every rank and for every slice of a sharded batch.  'time': the round-3
        # behaviour, an explicit opt-in -- time both once per shape on the first eager call and keep
        # the faster (the result can differ from run to run on near ties).  `last_path` says which ran.
        self.fused = fused
        self._auto_choice = {}
        self.last_path = None
        # derived weight tensors, one slot per (what, feature dtype, number of leading points kept):
        # slot -> (version key of conv_final's parameters, tensor(s)).  Several slots live side by side
        # (an f32 and an f16 pack; the full head and its latent prefix): a HIP graph captured over one
        # of them reads it by address (graph_cache.BatchGraph keeps `packed_snapshot()` alive and
        # re-checks it before every replay).
        self._derived = {}

    def _version_key(self):
        w, b = self.conv_final.weight, self.conv_final.bias
        if w.is_inference() or b.is_inference():   # (no version counters: see _trackable)
            return (w.data_ptr(), None, b.data_ptr(), None, w.device)
        return (w.data_ptr(), w._version, b.data_ptr(), b._version, w.device)

    def _trackable(self):
        # in-place parameter updates are seen through the version counters; inference tensors
        # (parameters created under torch.inference_mode) have none, so what derives from them is
        # rebuilt per call (one or two tiny launches; inside a capture: part of the graph)
        return not (self.conv_final.weight.is_inference() or self.conv_final.bias.is_inference())

    def _point_rows(self, n_keep, device):
        """Output channels of the first n_keep points: [j < n_keep] of the 2D block and of every depth
        slice of the 3D block (channel J + d*J + j, models/metrabs.py:79)."""
        J, D = self.n_points, self.config.depth
        j = torch.arange(n_keep, device=device)
        return torch.cat([j] + [J + d * J + j for d in range(D)])

    def _weights(self, n_keep=None):
        """(weight [n_out, C], bias [n_out]) of the whole head, or of its first n_keep points."""
        w, b = self.conv_final.weight, self.conv_final.bias
        w2 = w.detach().reshape(w.shape[0], -1)
        if n_keep is None or n_keep == self.n_points:
            return w2, b.detach()
        slot = ('rows', None, n_keep)
        key = self._version_key()
        hit = self._derived.get(slot)
        if not self._trackable() or hit is None or hit[0] != key:
            rows = self._point_rows(n_keep, w.device)
            hit = (key, (w2.index_select(0, rows).contiguous(), b.detach().index_select(0, rows).contiguous()))
            self._derived[slot] = hit
        return hit[1]

    def _packed_weights(self, feat_dtype, n_keep=None):
        n_keep = self.n_points if n_keep is None else int(n_keep)
        slot = ('packed', feat_dtype, n_keep)
        key = self._version_key()
        hit = self._derived.get(slot)
        if not self._trackable() or hit is None or hit[0] != key:
            w2, b = self._weights(n_keep)
            hit = (key, kernels.head_pack_weights(w2, b, n_keep, self.config.depth, feat_dtype))
            self._derived[slot] = hit
        return hit[1]

    def packed_snapshot(self):
        """What a captured graph may have read by address: {slot: (version key, tensors)} (references
        keep the memory from being handed out again while the graph lives)."""
        return dict(self._derived)

    def snapshot_is_current(self, snapshot):
        """False once conv_final's parameters were edited in place, replaced or moved after `snapshot`
        was taken, or a slot of it was rebuilt (its tensor is no longer the one the graph reads)."""
        key = self._version_key()
        trackable = self._trackable()
        for slot, (k, tensors) in snapshot.items():
            if k != key:
                return False
            # (untrackable parameters: every call -- a capture included -- builds its own derived tensors, so a
            #  graph reads the ones made inside ITS capture, whatever the slot holds now)
            cur = self._derived.get(slot)
            if trackable and (cur is None or cur[1] is not tensors):
                return False
        return True

    def forward(self, inp, first_points=None):
        """first_points = k: only the first k points are computed (their weight rows are selected once
        per weight version; every point's soft-argmax is independent of the others, so the result is
        what `forward(inp)[...][:, :k]` gives -- Metrabs' predict_all_and_latents slicing,
        models/metrabs.py:52-54 -- without computing or storing the rest)."""
        if isinstance(self.conv_final, torch.nn.modules.lazy.LazyModuleMixin) and \
                self.conv_final.has_uninitialized_params():
            self.conv_final(inp[:1])  # materialise the lazy conv exactly like the reference would
        n_keep = self.n_points if first_points is None else int(first_points)
        if not 0 < n_keep <= self.n_points:
            raise ValueError(f'first_points = {first_points} of {self.n_points} points')
        _, c_in, h, w = inp.shape
        use_fused = bool(self.fused) and kernels.head_fused_supported(
            c_in, n_keep, self.config.depth, h, w, kernels._is_channels_last(inp), inp.dtype)
        if use_fused and self.fused == 'auto':
            use_fused = kernels.head_auto_choice(c_in, n_keep, self.config.depth, h, w,
                                                 kernels._is_channels_last(inp), inp.dtype)
            self._auto_choice[(tuple(inp.shape[1:]), inp.dtype, kernels._is_channels_last(inp))] = use_fused
        elif use_fused and self.fused == 'time':
            use_fused = self._timed_pick(inp, n_keep)
        self.last_path = 'fused' if use_fused else 'library'  # (bench.py reports which one ran)
        if use_fused:
            return self._forward_fused(inp, n_keep)
        return self._forward_unfused(inp, n_keep)

    def _forward_fused(self, inp, n_keep=None):
        n_keep = self.n_points if n_keep is None else n_keep
        return kernels.head_fused(inp, self._packed_weights(inp.dtype, n_keep), inp.shape[1], n_keep,
                                  self.config)

    def _timed_pick(self, inp, n_keep=None):
        key = (tuple(inp.shape), inp.dtype, kernels._is_channels_last(inp)) + \
            (() if n_keep in (None, self.n_points) else (n_keep,))
        if key not in self._auto_choice:
            if torch.cuda.is_current_stream_capturing():
                return True  # nothing can be timed inside a capture; decided on an eager call
            fns = (lambda x: self._forward_fused(x, n_keep), lambda x: self._forward_unfused(x, n_keep))
            for fn in fns:  # lazy initialisation (weight packing, MIOpen's solver search)
                for _ in range(3):
                    fn(inp)
            best = [float('inf'), float('inf')]
            for _ in range(3):  # interleaved rounds, the minimum of each path (clock ramps, other streams)
                for i, fn in enumerate(fns):
                    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    start.record()
                    for _ in range(20):
                        fn(inp)
                    stop.record()
                    stop.synchronize()
                    best[i] = min(best[i], start.elapsed_time(stop))
            self._auto_choice[key] = best[0] <= best[1]
        return self._auto_choice[key]

    def _forward_unfused(self, inp, n_keep=None):
        # 1x1 conv as a library GEMM (rocBLAS / MIOpen).  16-bit features (the autocast backbone's
        # output) meet f32 parameters here: run the conv as autocast would (multiperson_model.py:241)
        n_keep = self.n_points if n_keep is None else n_keep
        if kernels._is_channels_last(inp):
            # channels_last features ARE the [B H W, C] matrix of a GEMM: F.linear on that view, NHWC logits (decoded in
            # place by the NHWC kernels) -- the library's channels_last 1x1 convolution is 2 - 5 x slower (round 6:
            # 64 crops of 8x8x1280 -> 1,241 channels 134 vs 315 us f32, 47 vs 192 us f16; profiles/r06x_layout_rule2.jsonl)
            w2, b = self._weights(n_keep)

            def conv(x):
                n, c, h, w = x.shape
                lg = torch.nn.functional.linear(x.permute(0, 2, 3, 1).reshape(n * h * w, c), w2, b)
                return lg.view(n, h, w, -1).permute(0, 3, 1, 2)
        elif n_keep == self.n_points:
            conv = self.conv_final
        else:
            w2, b = self._weights(n_keep)
            conv = lambda x: torch.nn.functional.conv2d(x, w2[:, :, None, None], b)
        if inp.dtype != self.conv_final.weight.dtype and inp.dtype in (torch.float16, torch.bfloat16):
            with torch.autocast(device_type=inp.device.type, dtype=inp.dtype):
                logits = conv(inp)
        else:
            logits = conv(inp)
        return kernels.softargmax_decode(logits, n_keep, self.config)


def load_affine_weights(spec):
    """FLAGS.affine_weights (models/metrabs.py:23-32): a path to an .npz holding `w1` [J, n_latents]
    (joints -> latent points) and `w2` [n_latents, J] (latent points -> joints), or a bare name looked
    up as $DATA_ROOT/skeleton_conversion/<name>.npz (posepile.paths.DATA_ROOT is the DATA_ROOT
    environment variable); a dict / npz object with those two arrays is taken as is."""
    if isinstance(spec, (str, os.PathLike)):
        path = os.fspath(spec)
        if not os.path.exists(path):
            path = os.path.join(os.environ.get('DATA_ROOT', ''), 'skeleton_conversion', f'{path}.npz')
        spec = np.load(path)
    w1 = torch.as_tensor(np.asarray(spec['w1']), dtype=torch.float32)
    w2 = torch.as_tensor(np.asarray(spec['w2']), dtype=torch.float32)
    if w1.dim() != 2 or w2.dim() != 2 or w1.shape != (w2.shape[1], w2.shape[0]):
        raise ValueError(f'affine weights: w1 {tuple(w1.shape)} and w2 {tuple(w2.shape)} are not a '
                         f'[J, n_latents] / [n_latents, J] pair')
    return w1, w2


class _deterministic_convolutions:
    """torch.backends.cudnn.deterministic = True for as long as ANY thread of the process is inside a pinned
    backbone call.  The flag is process-global: a plain `cudnn.flags(...)` per forward lets thread A's exit
    switch it off under thread B's running backbone.  A lock-protected depth counter: the first
    pinned call to enter saves the caller's setting and sets the flag, the last one to leave restores it; a
    user who set the flag themselves keeps it (restore puts back what was found)."""
    _lock = threading.Lock()
    _depth = 0
    _found = False

    def __enter__(self):
        cls = _deterministic_convolutions
        with cls._lock:
            if cls._depth == 0:
                cls._found = torch.backends.cudnn.deterministic
                torch.backends.cudnn.deterministic = True
            cls._depth += 1

    def __exit__(self, *exc):
        cls = _deterministic_convolutions
        with cls._lock:
            cls._depth -= 1
            if cls._depth == 0:
                torch.backends.cudnn.deterministic = cls._found
        return False


class Metrabs(torch.nn.Module):
    """models/metrabs.py:15-64 (TF twin metrabs_tf/models/metrabs.py:16-87), the affine-latent options
    included: with ``affine_weights`` the head predicts ``n_latents`` latent points
    (``transform_coords``), the latent points followed by the joints (``predict_all_and_latents``), or
    the joints alone (``regularize_to_manifold``, a training-time loss only); forward reconstructs the
    latent points and maps them to the joints with `recombination_weights` (latent_points_to_joints --
    the PyTorch file calls it at :62 without defining it; the TF file's :80-81 is the definition)."""

    def __init__(self, backbone, joint_info, config=None, in_channels=None, fused_head='auto',
                 autocast_dtype=None, affine_weights=None):
        super().__init__()
        # The reference runs the crop model under torch.autocast(float16) on the GPU
        # (multiperson_model.py:241); None keeps the fp32 arithmetic of its CPU path.
        self.autocast_dtype = autocast_dtype
        self.config = MetrabsConfig.from_any(config) if config is not None else MetrabsConfig()
        self.backbone = backbone
        self.joint_names = np.array(joint_info.names)
        self.joint_edges = np.array([[i, j] for i, j in joint_info.stick_figure_edges])
        self.input_resolution = np.int32(self.config.proc_side)
        self.joint_info = joint_info
        cfg = self.config
        affine = affine_weights if affine_weights is not None else cfg.affine_weights
        self.n_latents = None
        self.latent_output = False   # forward ends with latent_points_to_joints
        self.latent_prefix = None    # forward keeps the first n_latents raw points
        if affine is not None and not (isinstance(affine, str) and not affine):
            w1, w2 = load_affine_weights(affine)
            if w2.shape[1] != joint_info.n_joints:
                raise ValueError(f'affine weights map to {w2.shape[1]} joints, the model has '
                                 f'{joint_info.n_joints}')
            self.n_latents = int(w2.shape[0])
            # (plain attributes in the reference: not part of its state_dict -> non-persistent buffers)
            self.register_buffer('recombination_weights', w2, persistent=False)
            self.register_buffer('encoder_weights', w1, persistent=False)
            self.register_buffer('reconstruction_weights', w1 @ w2, persistent=False)
            if cfg.transform_coords:
                n_raw_points = self.n_latents
            elif cfg.predict_all_and_latents:
                n_raw_points = self.n_latents + joint_info.n_joints
            elif cfg.regularize_to_manifold:
                n_raw_points = joint_info.n_joints
            else:
                raise ValueError('affine weights not used')  # (models/metrabs.py:41)
            self.latent_output = bool(cfg.transform_coords or cfg.predict_all_and_latents)
            # models/metrabs.py:52-54 slices under predict_all_and_latents alone; under transform_coords
            # every raw point is a latent point already
            if cfg.predict_all_and_latents:
                self.latent_prefix = self.n_latents
        else:
            if cfg.transform_coords or cfg.predict_all_and_latents:
                # the reference would fail at forward time (no recombination_weights); fail at construction
                raise ValueError('transform_coords / predict_all_and_latents need affine_weights')
            n_raw_points = joint_info.n_joints
        self.n_raw_points = n_raw_points
        self.heatmap_heads = MetrabsHeads(
            n_points=n_raw_points, config=self.config, in_channels=in_channels, fused=fused_head)
        # set by Pose3dEstimator(shard_across_ranks='exact_monolithic') around its calls
        self.exact_monolithic = False
        # bumped whenever the module's tensors are moved / cast (nn.Module._apply): graphs captured over
        # the old storage are dropped by their owner (graph_cache.GraphCache.plan_call)
        self.storage_generation = 0

    def _apply(self, fn, *args, **kwargs):
        self.storage_generation = getattr(self, 'storage_generation', 0) + 1
        return super()._apply(fn, *args, **kwargs)

    def latent_points_to_joints(self, points):
        """metrabs_tf/models/metrabs.py:80-81 (tfu3d.linear_combine_points, tfu3d.py:48-49)."""
        return kernels.linear_combine_points(points, self.recombination_weights)

    def joints_to_latent_points(self, points):
        """metrabs_tf/models/metrabs.py:83-84."""
        return kernels.linear_combine_points(points, self.encoder_weights)

    def joints_to_joints(self, points):
        """metrabs_tf/models/metrabs.py:86-87."""
        return kernels.linear_combine_points(points, self.reconstruction_weights)

    # The backbone is PyTorch-ROCm's (MIOpen / rocBLAS): with PyTorch's default settings MIOpen may pick
    # solvers that accumulate with atomics, and the SAME call on the SAME input then differs from run to run
    # (measured on one MI355X box, EfficientNetV2-S, 64 crops: features 7e-6 apart in f32, 7e-2 under f16
    # autocast; poses 1e-2 mm / several mm; on another box 0.0 -- it depends on the box's MIOpen state:
    # tools/experiments/backbone_determinism_probe.py, profiles/r05f_ / r05z_backbone_determinism.jsonl).
    # True: the backbone runs under torch.backends.cudnn.flags(deterministic=True) -- eager calls, captured
    # graphs, replays and module copies then agree bit for bit, which is what lets a replayed HIP graph be "the
    # eager path's bits".  False: PyTorch's global setting decides.  None (default): pinned for f32 arithmetic --
    # the parity target, where the pin costs 0.6 - 2.2 % of the step (EfficientNetV2-S; 5 % at EfficientNetV2-L) -- and not under 16-bit autocast, whose own
    # rounding (1.8 mm mean from the f32 model) is the size of the run-to-run noise and where the pin costs up
    # to 13 % (EfficientNetV2-L 384 f16: 1.86 k -> 1.62 k crops/s, profiles/r05z_bench_config4.json).
    deterministic_backbone = None

    def backbone_is_pinned(self):
        d = self.deterministic_backbone
        return bool(self.autocast_dtype is None if d is None else d)

    def _run_backbone(self, image):
        if self.backbone_is_pinned() and image.is_cuda:
            with _deterministic_convolutions():
                return self.backbone(image)
        return self.backbone(image)

    def predict_multi(self, image, intrinsic_matrix):
        """The bare-bones crop-model entry of the TF twin (metrabs_tf/models/metrabs.py:71-78,
        docs/INFERENCE.md:112-132): ``image`` float16 [N, res, res, 3] -- crops the caller made itself,
        principal point at the crop centre --, ``intrinsic_matrix`` float32 [N, 3, 3] -> poses3d float32
        [N, J, 3] in the crop camera's frame (mm).  The TF signature is strict about both dtypes and so is
        this one.  The interleaved crops are handed to the backbone as the NCHW VIEW of the same memory
        (torch channels_last: no layout copy), the crop model runs under 16-bit autocast (the exported TF
        model's mixed_float16 policy; `autocast_dtype` if the model names one) and the head consumes the
        NHWC 16-bit features in place."""
        if image.dtype != torch.float16 or image.dim() != 4 or image.shape[-1] != 3:
            raise TypeError(f'predict_multi: image must be float16 [N, H, W, 3], got {image.dtype} '
                            f'{tuple(image.shape)}')
        if intrinsic_matrix.dtype != torch.float32 or tuple(intrinsic_matrix.shape) != (image.shape[0], 3, 3):
            raise TypeError(f'predict_multi: intrinsic_matrix must be float32 [{image.shape[0]}, 3, 3], got '
                            f'{intrinsic_matrix.dtype} {tuple(intrinsic_matrix.shape)}')
        kernels.require_cuda(image, intrinsic_matrix)
        crops = image.contiguous().permute(0, 3, 1, 2)   # [N, 3, H, W] over NHWC memory
        return self.forward((crops, intrinsic_matrix), autocast_dtype=self.autocast_dtype or torch.float16)

    def forward(self, inp, autocast_dtype=None):
        image, intrinsics = inp
        autocast_dtype = self.autocast_dtype if autocast_dtype is None else autocast_dtype
        if autocast_dtype is not None:
            with torch.autocast('cuda', dtype=autocast_dtype):
                features = self._run_backbone(image)
        else:
            features = self._run_backbone(image)
        # predict_all_and_latents: coords[:, :n_latents] (models/metrabs.py:52-54) -- the head computes
        # just those points
        coords2d, coords3d = self.heatmap_heads(features, first_points=self.latent_prefix)
        if self.exact_monolithic and distributed.exact_mode_needs_allreduce(self):
            # this call holds one rank's slice of a reference internal batch: the batch-global RMS
            # scalars of reconstruct_ref_fullpersp (ptu3d.py:71-74) come from the summed moments
            moments = kernels.reconstruct_moments(coords2d, coords3d, intrinsics)
            distributed.allreduce_moments(moments)
            coords3d_abs = kernels.reconstruct_solve(coords2d, coords3d, intrinsics, moments, self.config)
        else:
            coords3d_abs = kernels.reconstruct_absolute(coords2d, coords3d, intrinsics, self.config)
        if self.latent_output:  # models/metrabs.py:61-62
            coords3d_abs = self.latent_points_to_joints(coords3d_abs)
        return coords3d_abs
