"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain-PyTorch fp32 CPU restatement of the MeTRAbs per-crop hot path of the reference
(isarandi/metrabs, ``metrabs_pytorch/``).  Every function cites the reference file:line it
follows.  Op order is kept wherever it decides rounding (softmax as exp(x-max)/sum, marginal sums
then dot with linspace, batch-global RMS, weights mask+1e-4, ridge rows) so that this file is
BIT-IDENTICAL to the reference on CPU; ``tests/test_oracle_pin.py`` proves that against the live
reference in the build container and against the golden vectors in ``tests/golden/`` everywhere.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this
file.  The product (``metrabs_amd``) never does: it fails loudly without its HIP library.

Differences from the reference that are deliberate (none changes a number):
  * configuration is an explicit ``HeadConfig`` instead of the global hydra ``get_config()``
    (metrabs_pytorch/util.py:41-57);
  * ragged splits use Python ints (the reference passes Tensors to torch.split, which torch 2.10
    rejects -- multiperson_model.py:154-155,171).
"""
import dataclasses
import math

import numpy as np
import torch
import torch.nn.functional as F


@dataclasses.dataclass
class HeadConfig:
    """Keys of metrabs_pytorch/config/config.yaml:1-22 (+ config_s_256.yaml:5-9) that the hot
    path reads at call time."""
    proc_side: int = 256
    stride_train: int = 32
    stride_test: int = 32
    centered_stride: bool = True
    legacy_centered_stride_bug: bool = False
    depth: int = 8
    box_size_mm: float = 2200.0
    weak_perspective: bool = False
    mix_3d_inside_fov: float = 0.5
    # the affine-latent options of Metrabs (models/metrabs.py:23-44,52-62); the weights themselves
    # (FLAGS.affine_weights names a file) travel beside the config
    transform_coords: bool = False
    predict_all_and_latents: bool = False
    regularize_to_manifold: bool = False

    def as_dict(self):
        return dataclasses.asdict(self)


# ----------------------------------------------------------------------------------------------
# ptu.py
# ----------------------------------------------------------------------------------------------

def ref_linspace(start, stop, num, dtype=None, endpoint=True):
    """ptu.linspace, metrabs_pytorch/ptu.py:78-92.  num==1 with endpoint=True gives the midpoint."""
    start = torch.as_tensor(start, dtype=dtype)
    stop = torch.as_tensor(stop, dtype=dtype)
    if endpoint:
        if num == 1:
            return torch.mean(torch.stack([start, stop], dim=0), dim=0, keepdim=True)
        return torch.linspace(start, stop, num, dtype=dtype)
    if num > 1:
        step = (stop - start) / num
        return torch.linspace(start, stop - step, num, dtype=dtype)
    return torch.linspace(start, stop, num, dtype=dtype)


def joint_softmax(x, dims):
    """ptu.softmax, metrabs_pytorch/ptu.py:47-51: one softmax jointly over several dims."""
    peak = torch.amax(x, dim=dims, keepdim=True)
    e = torch.exp(x - peak)
    return e / torch.sum(e, dim=dims, keepdim=True)


def expectation_decode(prob, dims):
    """ptu.decode_heatmap, metrabs_pytorch/ptu.py:58-75.  For every heatmap axis: marginalise over
    the other heatmap axes, then dot with linspace(0, 1, n).  Output coordinate order = ``dims``
    order."""
    dims = tuple(d if d >= 0 else prob.ndim + d for d in dims)
    coords = []
    for d in dims:
        others = [o for o in dims if o != d]
        marginal = torch.sum(prob, dim=others, keepdim=True)
        grid = ref_linspace(0.0, 1.0, prob.shape[d], dtype=prob.dtype)
        val = torch.tensordot(marginal, grid, dims=([d], [0]))
        val = torch.unsqueeze(val, d)
        for hd in sorted(dims, reverse=True):
            val = val.squeeze(hd)
        coords.append(val)
    return torch.stack(coords, dim=-1)


def soft_argmax(x, dims):
    """ptu.soft_argmax, metrabs_pytorch/ptu.py:54-55."""
    return expectation_decode(joint_softmax(x, dims), dims)


def reduce_sum_masked(x, valid, dim, keepdim):
    """ptu.reduce_sum_masked, metrabs_pytorch/ptu.py:37-44 (dim given)."""
    valid = valid.reshape(tuple(valid.shape) + (1,) * (x.ndim - valid.ndim))
    return torch.where(valid, x, torch.zeros_like(x)).sum(dim=dim, keepdim=keepdim)


def reduce_mean_masked(x, valid, dim, keepdim):
    """ptu.reduce_mean_masked, metrabs_pytorch/ptu.py:22-34 (mask and dim given)."""
    valid = valid.reshape(tuple(valid.shape) + (1,) * (x.ndim - valid.ndim))
    total = torch.where(valid, x, torch.zeros_like(x)).sum(dim=dim, keepdim=keepdim)
    count = valid.sum(dim=dim, keepdim=keepdim, dtype=x.dtype)
    return torch.nan_to_num(total / count)


def mean_stdev_masked(x, valid, items_dim, dimensions_dim):
    """ptu.mean_stdev_masked, metrabs_pytorch/ptu.py:4-19 (fixed_ref=None)."""
    mean = reduce_mean_masked(x, valid, dim=items_dim, keepdim=True)
    centered = x - mean
    valid_b = valid.reshape(tuple(valid.shape) + (1,) * (x.ndim - valid.ndim))
    n_valid = valid_b.sum(dim=items_dim, keepdim=True, dtype=x.dtype)
    ssd = reduce_sum_masked(
        torch.square(centered), valid_b, dim=(items_dim, dimensions_dim), keepdim=True)
    stdev = torch.sqrt(torch.nan_to_num(ssd / n_valid) + 1e-10)
    return mean, stdev


# ----------------------------------------------------------------------------------------------
# models/util.py and models/metrabs.py (MetrabsHeads)
# ----------------------------------------------------------------------------------------------

def heatmap_to_image(coords, cfg, is_training=False):
    """models/util.py:6-20."""
    stride = cfg.stride_train if is_training else cfg.stride_test
    last_pixel = cfg.proc_side - 1
    last_center = last_pixel - (last_pixel % stride)
    out = coords * last_center
    if cfg.centered_stride:
        out = out + stride // 2
    if cfg.legacy_centered_stride_bug:
        out = out + stride // 2
    return out


def heatmap_to_metric(coords, cfg, is_training=False):
    """models/util.py:29-33."""
    xy = heatmap_to_image(coords[..., :2], cfg, is_training) * cfg.box_size_mm / cfg.proc_side
    return torch.cat([xy, coords[..., 2:] * cfg.box_size_mm], dim=-1)


def heads_from_logits(logits, n_points, cfg, eval_dtype=torch.float32):
    """MetrabsHeads.forward after the 1x1 conv, models/metrabs.py:78-85.

    logits: [B, n_points*(1+depth), H, W]; channel n<J is the 2D heatmap of joint n, channel
    J + d*J + j is depth slice d of joint j ('b (d j) h w -> b d j h w').
    eval_dtype=float64 evaluates the same formulas in double: the yardstick that tells the
    reference's own fp32 rounding from ours (never the parity target)."""
    j = n_points
    logits2d, logits3d = torch.split(logits, [j, cfg.depth * j], dim=1)
    b, _, h, w = logits3d.shape
    logits3d = logits3d.reshape(b, cfg.depth, j, h, w)
    coords3d = soft_argmax(logits3d.to(eval_dtype), dims=(4, 3, 1))
    coords3d_rel = heatmap_to_metric(coords3d, cfg)
    coords2d = soft_argmax(logits2d.to(eval_dtype), dims=(3, 2))
    coords2d_px = heatmap_to_image(coords2d, cfg)
    return coords2d_px, coords3d_rel


def heads_forward(features, weight, bias, n_points, cfg):
    """MetrabsHeads.forward, models/metrabs.py:75-85.  weight: [J(1+D), C, 1, 1], bias [J(1+D)]
    (torch.nn.LazyConv2d(kernel_size=1), models/metrabs.py:73)."""
    if weight.ndim == 2:
        weight = weight[:, :, None, None]
    logits = F.conv2d(features, weight, bias)
    return heads_from_logits(logits, n_points, cfg)


# ----------------------------------------------------------------------------------------------
# ptu3d.py
# ----------------------------------------------------------------------------------------------

def to_homogeneous(x):
    """ptu3d.py:52-53."""
    return torch.cat([x, torch.ones_like(x[..., :1])], dim=-1)


def project(points):
    """ptu3d.py:145-146."""
    return points[..., :2] / points[..., 2:3]


def back_project(camcoords2d, delta_z, z_offset):
    """ptu3d.py:108-110."""
    return to_homogeneous(camcoords2d) * torch.unsqueeze(
        delta_z + torch.unsqueeze(z_offset, -1), -1)


def is_within_fov(imcoords, cfg, border_factor=0.75):
    """ptu3d.py:113-121."""
    offset = -cfg.stride_train / 2 if not cfg.centered_stride else 0
    lower = cfg.stride_train * border_factor + offset
    upper = cfg.proc_side - cfg.stride_train * border_factor + offset
    return torch.all(torch.logical_and(imcoords >= lower, imcoords <= upper), dim=-1)


def reconstruct_ref_weakpersp(normalized_2d, coords3d_rel, validity_mask):
    """ptu3d.py:36-49."""
    mean3d, stdev3d = mean_stdev_masked(
        coords3d_rel[..., :2], validity_mask, items_dim=1, dimensions_dim=2)
    mean2d, stdev2d = mean_stdev_masked(
        normalized_2d[..., :2], validity_mask, items_dim=1, dimensions_dim=2)
    stdev2d = torch.maximum(stdev2d, torch.tensor(1e-5))
    stdev3d = torch.maximum(stdev3d, torch.tensor(1e-5))
    old_mean = reduce_mean_masked(coords3d_rel, validity_mask, dim=1, keepdim=True)
    new_mean_z = torch.nan_to_num(stdev3d / stdev2d)
    new_mean = to_homogeneous(mean2d) * new_mean_z
    return torch.squeeze(new_mean - old_mean, 1)


def _batch_rms_normalize(x):
    """rms_normalize inside reconstruct_ref_fullpersp, ptu3d.py:71-74.  NB: the RMS is over the
    WHOLE tensor, i.e. over every crop of the crop_model call."""
    scale = x.square().mean().sqrt()
    return scale, x / scale


def reconstruct_ref_fullpersp(normalized_2d, coords3d_rel, validity_mask):
    """ptu3d.py:56-105: weighted ridge least squares for the reference point.

    Rows per joint: [1 0 -x; 0 1 -y] ref = x*z_rel - xy_rel, with x and the rhs each divided by a
    batch-global RMS; weights mask+1e-4; three ridge rows sqrt(1e-2)*I; torch.linalg.lstsq."""
    n_batch, n_points = normalized_2d.shape[:2]
    dtype = normalized_2d.dtype
    eye2 = torch.eye(2, dtype=dtype).unsqueeze(0).repeat(n_batch, n_points, 1)
    scale2d, xy_n = _batch_rms_normalize(normalized_2d.reshape(-1, n_points * 2, 1))
    a_data = torch.cat([eye2, -xy_n], dim=2)
    eye3 = torch.eye(3, dtype=dtype).unsqueeze(0).repeat(n_batch, 1, 1)
    a_full = torch.cat([a_data, eye3], dim=1)

    rel_backproj = normalized_2d * coords3d_rel[:, :, 2:] - coords3d_rel[:, :, :2]
    scale_rhs, rhs = _batch_rms_normalize(rel_backproj.reshape(-1, n_points * 2, 1))
    rhs_full = torch.cat([rhs, torch.zeros(n_batch, 3, 1, dtype=torch.float32)], dim=1)

    w = validity_mask.float() + np.float32(1e-4)
    w = torch.repeat_interleave(w, 2, dim=1).unsqueeze(-1)  # 'b j -> b (j c) 1', c=2
    w_ridge = torch.full((n_batch, 3, 1), np.sqrt(1e-2), dtype=torch.float32)
    w_full = torch.cat([w, w_ridge], dim=1)

    ref = torch.linalg.lstsq(a_full * w_full, rhs_full * w_full).solution
    ref = torch.cat([ref[:, :2] * scale_rhs, ref[:, 2:] * (scale_rhs / scale2d)], dim=1)
    return torch.squeeze(ref, dim=-1)


def reconstruct_absolute(coords2d, coords3d_rel, intrinsics, cfg, mix_3d_inside_fov='cfg',
                         weak_perspective=None):
    """ptu3d.py:9-33.  ``mix_3d_inside_fov='cfg'`` takes cfg.mix_3d_inside_fov, which is what
    Metrabs.forward passes (models/metrabs.py:57-59)."""
    if mix_3d_inside_fov == 'cfg':
        mix_3d_inside_fov = cfg.mix_3d_inside_fov
    inv_k = torch.linalg.inv(intrinsics.to(coords2d.dtype))
    normalized = (to_homogeneous(coords2d) @ inv_k.transpose(1, 2))[..., :2]
    if weak_perspective is None:
        weak_perspective = cfg.weak_perspective
    in_fov = is_within_fov(coords2d, cfg)
    if weak_perspective:
        ref = reconstruct_ref_weakpersp(normalized, coords3d_rel, in_fov)
    else:
        ref = reconstruct_ref_fullpersp(normalized, coords3d_rel, in_fov)
    abs_3d_based = coords3d_rel + ref[:, np.newaxis]
    abs_2d_based = back_project(normalized, coords3d_rel[..., 2], ref[:, 2])
    if mix_3d_inside_fov is not None:
        abs_2d_based = (mix_3d_inside_fov * abs_3d_based +
                        (1 - mix_3d_inside_fov) * abs_2d_based)
    return torch.where(in_fov[..., np.newaxis], abs_2d_based, abs_3d_based)


def linear_combine_points(coords, weights):
    """tfu3d.linear_combine_points, metrabs_tf/tfu3d.py:48-49 (the PyTorch port calls
    Metrabs.latent_points_to_joints at models/metrabs.py:62 without defining it; the TF twin's
    definition, metrabs_tf/models/metrabs.py:80-81, is the specification)."""
    return torch.einsum('bjc,jJ->bJc', coords, weights)


def n_raw_points(n_joints, n_latents, cfg):
    """Metrabs.__init__, models/metrabs.py:34-44: how many points the head predicts."""
    if n_latents is None:
        return n_joints
    if cfg.transform_coords:
        return n_latents
    if cfg.predict_all_and_latents:
        return n_latents + n_joints
    if cfg.regularize_to_manifold:
        return n_joints
    raise Exception('affine weights not used')


def crop_model_from_features(features, weight, bias, intrinsics, n_points, cfg, recombination_weights=None):
    """Metrabs.forward after the backbone, models/metrabs.py:50-64.  n_points = the head's raw point
    count; recombination_weights [n_latents, J] (`w2` of the affine-weights file) when the model was
    built with affine weights."""
    coords2d, coords3d_rel = heads_forward(features, weight, bias, n_points, cfg)
    if cfg.predict_all_and_latents:  # :52-54
        n_latents = recombination_weights.shape[0]
        coords2d = coords2d[:, :n_latents]
        coords3d_rel = coords3d_rel[:, :n_latents]
    coords3d_abs = reconstruct_absolute(coords2d, coords3d_rel, intrinsics, cfg)
    if cfg.transform_coords or cfg.predict_all_and_latents:  # :61-62
        coords3d_abs = linear_combine_points(coords3d_abs, recombination_weights)
    return coords3d_abs


def lookat_matrix(forward_vector, up_vector):
    """ptu3d.py:129-142."""
    new_z = forward_vector / torch.linalg.norm(forward_vector, dim=-1, keepdim=True)
    new_x = torch.linalg.cross(new_z, up_vector)
    new_x_alt = torch.stack([new_z[:, 2], torch.zeros_like(new_z[:, 2]), -new_z[:, 0]], dim=1)
    new_x = torch.where(torch.linalg.norm(new_x, dim=-1, keepdim=True) == 0, new_x_alt, new_x)
    new_x = new_x / torch.linalg.norm(new_x, dim=-1, keepdim=True)
    new_y = torch.linalg.cross(new_z, new_x)
    return torch.stack([new_x, new_y, new_z], dim=1)


def intrinsic_matrix_from_field_of_view(fov_degrees, imshape):
    """ptu3d.py:149-161."""
    imshape = torch.tensor(imshape, dtype=torch.float32)
    fov_radians = fov_degrees * torch.tensor(np.pi / 180, dtype=torch.float32)
    larger_side = torch.max(imshape)
    focal = larger_side / (torch.tan(fov_radians / 2) * 2)
    zero = torch.tensor(0, dtype=torch.float32)
    one = torch.tensor(1, dtype=torch.float32)
    return torch.stack([
        torch.stack([focal, zero, imshape[1] / 2], dim=-1),
        torch.stack([zero, focal, imshape[0] / 2], dim=-1),
        torch.stack([zero, zero, one], dim=-1)], dim=-2).unsqueeze(0)


def rotation_mat_z(angle):
    """ptu3d.rotation_mat(angle, rot_axis='z'), ptu3d.py:164-184."""
    sin, cos = torch.sin(angle), torch.cos(angle)
    zero, one = torch.zeros_like(angle), torch.ones_like(angle)
    return torch.stack([
        torch.stack([cos, -sin, zero], dim=-1),
        torch.stack([sin, cos, zero], dim=-1),
        torch.stack([zero, zero, one], dim=-1)], dim=-2)


# ----------------------------------------------------------------------------------------------
# multiperson/warping.py
# ----------------------------------------------------------------------------------------------

def corner_aligned_scale_mat(factor):
    """warping.py:128-133."""
    shift = (factor - 1) / 2
    return torch.from_numpy(np.array(
        [[factor, 0, shift], [0, factor, shift], [0, 0, 1]], dtype=np.float32))


def _pad_last_to(x, size):
    """warping.pad_axis_to_size(x, size, -1), warping.py:110-113."""
    return F.pad(x, (0, size - x.shape[-1]))


def distortion_formula_parts(points, coeffs):
    """warping.py:90-107.  OpenCV order (k1,k2,p1,p2,k3,k4,k5,k6,s1,s2,s3,s4), zero padded."""
    d = _pad_last_to(coeffs, 12)
    shape = ([-1] if d.ndim > 1 else []) + [1] * (points.ndim - d.ndim) + [12]
    d = torch.reshape(d, shape)
    r2 = torch.sum(torch.square(points), dim=-1, keepdim=True)
    a = ((((d[..., 4:5] * r2 + d[..., 1:2]) * r2 + d[..., 0:1]) * r2 + 1) /
         (((d[..., 7:8] * r2 + d[..., 6:7]) * r2 + d[..., 5:6]) * r2 + 1))
    p2_1 = torch.flip(d[..., 2:4], dims=[-1])
    b = 2 * torch.sum(points * p2_1, dim=-1, keepdim=True)
    c = (d[..., 9:12:2] * r2 + p2_1 + d[..., 8:11:2]) * r2
    return a, b, c


def distort_points(points, coeffs):
    """warping.py:57-62 (all-zero coefficients short-circuit bit-exactly)."""
    if torch.all(coeffs == 0):
        return points
    a, b, c = distortion_formula_parts(points, coeffs)
    return points * (a + b) + c


def undistort_points(points, coeffs):
    """warping.py:65-73: five fixed-point iterations."""
    if torch.all(coeffs == 0):
        return points
    und = points
    for _ in range(5):
        a, b, c = distortion_formula_parts(und, coeffs)
        und = (points - c - und * b) / a
    return und


def warp_single_image(image, intrinsic_matrix, new_invprojmat, distortion_coeffs, output_shape,
                      eval_dtype=None):
    """warping.py:41-54.  image [3,H,W] linear light; returns [3,oh,ow].
    eval_dtype=torch.float64: the same formulas on the same (f32) matrices and texels in double --
    the yardstick for "whose rounding is it" of the sampler gates; the default is the reference's f32."""
    grid = torch.stack(torch.meshgrid(
        torch.arange(output_shape[1]), torch.arange(output_shape[0]), indexing='xy'),
        dim=-1).float()
    if eval_dtype is not None:
        image, intrinsic_matrix, new_invprojmat, distortion_coeffs, grid = (
            t.to(eval_dtype) for t in (image, intrinsic_matrix, new_invprojmat, distortion_coeffs, grid))
    rays = torch.einsum('hwc,Cc->hwC', to_homogeneous(grid), new_invprojmat)
    rays = to_homogeneous(distort_points(project(rays), distortion_coeffs))
    src = torch.einsum('hwc,Cc->hwC', rays, intrinsic_matrix)[..., :2]
    size = torch.tensor([image.shape[2], image.shape[1]], dtype=src.dtype)
    src_n = (src / (size - 1)) * 2 - 1
    return F.grid_sample(
        image.unsqueeze(0), src_n.unsqueeze(0), align_corners=True, mode='bilinear',
        padding_mode='zeros').squeeze(0)


def build_pyramid(images, n_levels=3):
    """warping.py:10-13: 2x2 box filter, floor on odd sizes."""
    levels = [images]
    for _ in range(1, n_levels):
        levels.append(F.avg_pool2d(levels[-1], 2, 2))
    return levels


def pyramid_level_index(crop_scales, n_levels=3):
    """warping.py:20-21."""
    return torch.clip(torch.floor(-torch.log2(crop_scales)), 0, n_levels - 1).int()


def warp_images_with_pyramid(images, intrinsic_matrix, new_invprojmats, distortion_coeffs,
                             crop_scales, output_shape, image_ids, n_pyramid_levels=3, eval_dtype=None):
    """warping.py:6-28."""
    levels = build_pyramid(images, n_pyramid_levels)
    k_levels = [corner_aligned_scale_mat(1 / 2 ** lvl) @ intrinsic_matrix
                for lvl in range(n_pyramid_levels)]
    lvl_idx = pyramid_level_index(crop_scales, n_pyramid_levels)
    return torch.stack([
        warp_single_image(
            levels[lvl_idx[i]][image_ids[i]], k_levels[lvl_idx[i]][i], new_invprojmats[i],
            distortion_coeffs[i], output_shape, eval_dtype=eval_dtype)
        for i in range(len(image_ids))])


# ----------------------------------------------------------------------------------------------
# multiperson/multiperson_model.py
# ----------------------------------------------------------------------------------------------

UNKNOWN_INTRINSIC_MATRIX = ((-1, -1, -1), (-1, -1, -1), (-1, -1, -1))
DEFAULT_EXTRINSIC_MATRIX = ((1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1))
DEFAULT_DISTORTION = (0, 0, 0, 0, 0)
DEFAULT_WORLD_UP = (0, -1, 0)


def tta_params(num_aug, rot_aug_degrees=25):
    """multiperson_model.py:108-137 -> dict(gammas, angles, scales, should_flip, rotflipmat)."""
    gammas = ref_linspace(np.float32(0.6), np.float32(1.0), num_aug)
    angle_range = np.float32(np.deg2rad(rot_aug_degrees))
    angles = ref_linspace(-angle_range, angle_range, num_aug)
    scales = torch.cat([
        ref_linspace(0.8, 1.0, num_aug // 2, endpoint=False),
        torch.linspace(1.0, 1.1, num_aug - num_aug // 2)], dim=0)
    should_flip = (torch.arange(0, num_aug) - num_aug // 2) % 2 != 0
    flipmat = torch.tensor([[-1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=torch.float32)
    maybe_flip = torch.where(should_flip[:, np.newaxis, np.newaxis], flipmat, torch.eye(3))
    rotflipmat = maybe_flip @ rotation_mat_z(-angles)
    return dict(gammas=gammas, angles=angles, scales=scales, should_flip=should_flip,
                rotflipmat=rotflipmat)


def get_new_rotation_and_scale(intrinsic_matrix, distortion_coeffs, camspace_up, boxes, res):
    """multiperson_model.py:322-355."""
    x, y, w, h = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    boxpoints = to_homogeneous(torch.stack([
        torch.stack([x + w / 2, y + h / 2], dim=1),
        torch.stack([x + w / 2, y], dim=1),
        torch.stack([x + w, y + h / 2], dim=1),
        torch.stack([x + w / 2, y + h], dim=1),
        torch.stack([x, y + h / 2], dim=1)], dim=1))
    cam = torch.einsum('bpc,bCc->bpC', boxpoints, torch.linalg.inv(intrinsic_matrix))
    cam = to_homogeneous(undistort_points(cam[:, :, :2], distortion_coeffs))
    r_noaug = lookat_matrix(forward_vector=cam[:, 0], up_vector=camspace_up)
    sides = project(torch.einsum('bpc,bCc->bpC', cam[:, 1:5], intrinsic_matrix @ r_noaug))
    vertical = torch.linalg.norm(sides[:, 0] - sides[:, 2], dim=-1)
    horizontal = torch.linalg.norm(sides[:, 1] - sides[:, 3], dim=-1)
    box_size = torch.maximum(vertical, horizontal)
    box_scales = torch.tensor(res, dtype=box_size.dtype) / box_size
    return r_noaug, box_scales


def get_crops(images_linear, intrinsic_matrix, distortion_coeffs, camspace_up, boxes, image_ids,
              aug_rotflipmat, aug_scales, aug_gammas, antialias_factor, res, eval_dtype=None):
    """multiperson_model.py:264-320.  images_linear: float linear-light [N,3,H,W].
    eval_dtype=torch.float64: the geometry (f32, as the reference computes it) is kept and the sampling
    itself -- coordinates, distortion, interpolation -- is evaluated in double (see warp_single_image)."""
    r_noaug, box_scales = get_new_rotation_and_scale(
        intrinsic_matrix, distortion_coeffs, camspace_up, boxes, res)
    crop_scales = aug_scales[:, np.newaxis] * box_scales[np.newaxis, :]
    num_box, num_aug = boxes.shape[0], aug_gammas.shape[0]
    new_k = torch.cat([
        torch.cat([
            intrinsic_matrix[np.newaxis, :, :2, :2] * crop_scales[:, :, np.newaxis, np.newaxis],
            torch.full((num_aug, num_box, 2, 1), res / 2, dtype=torch.float32)], dim=3),
        torch.cat([
            torch.zeros((num_aug, num_box, 1, 2), dtype=torch.float32),
            torch.ones((num_aug, num_box, 1, 1), dtype=torch.float32)], dim=3)], dim=2)
    rot = aug_rotflipmat[:, np.newaxis] @ r_noaug
    new_invprojmat = torch.linalg.inv(new_k @ rot)
    if antialias_factor > 1:
        new_invprojmat = new_invprojmat @ corner_aligned_scale_mat(1 / antialias_factor)
    crops = warp_images_with_pyramid(
        images_linear,
        intrinsic_matrix=torch.tile(intrinsic_matrix, [num_aug, 1, 1]),
        new_invprojmats=torch.reshape(new_invprojmat, [-1, 3, 3]),
        distortion_coeffs=torch.tile(distortion_coeffs, [num_aug, 1]),
        crop_scales=torch.reshape(crop_scales, [-1]) * antialias_factor,
        output_shape=(res * antialias_factor, res * antialias_factor),
        image_ids=torch.tile(image_ids, [num_aug]), eval_dtype=eval_dtype)
    if antialias_factor == 2:
        crops = F.avg_pool2d(crops, 2, 2)
    elif antialias_factor == 4:
        crops = F.avg_pool2d(crops, 4, 4)
    elif antialias_factor > 4:
        # torchvision.transforms.functional.resize(crops, (res, res), BILINEAR, antialias=True)
        # (multiperson_model.py:312-315); torchvision's tensor path is exactly this call
        # (transforms/_functional_tensor.py: resize -> torch.nn.functional.interpolate)
        crops = F.interpolate(crops, size=[res, res], mode='bilinear', align_corners=False, antialias=True)
    crops = torch.reshape(crops, [num_aug, num_box, 3, res, res])
    crops **= torch.reshape(aug_gammas / 2.2, [-1, 1, 1, 1, 1]).to(crops.dtype)
    return crops, new_k, rot


def predict_single_batch(crop_model, mirror_mapping, n_joints, images_linear, intrinsic_matrix,
                         distortion_coeffs, camspace_up, boxes, image_ids, tta, antialias_factor,
                         res):
    """multiperson_model.py:227-259.  crop_model((crops[n,3,res,res], K[n,3,3])) -> [n,J,3]."""
    crops, new_k, rot = get_crops(
        images_linear, intrinsic_matrix, distortion_coeffs, camspace_up, boxes, image_ids,
        tta['rotflipmat'], tta['scales'], tta['gammas'], antialias_factor, res)
    poses_flat = crop_model((torch.reshape(crops, (-1, 3, res, res)),
                             torch.reshape(new_k, (-1, 3, 3))))
    num_aug = tta['should_flip'].shape[0]
    poses = torch.reshape(poses_flat, [num_aug, -1, n_joints, 3])
    swapped = poses[..., mirror_mapping, :]
    poses = torch.where(torch.reshape(tta['should_flip'], [-1, 1, 1, 1]), swapped, poses)
    return (poses @ rot).transpose(0, 1)


def predict_in_batches(crop_model, mirror_mapping, n_joints, images_u8, intrinsic_matrix,
                       distortion_coeffs, camspace_up, boxes, internal_batch_size, tta,
                       antialias_factor, res):
    """multiperson_model.py:184-225."""
    num_aug = len(tta['gammas'])
    boxes_per_batch = internal_batch_size // num_aug
    boxes_flat = torch.cat(boxes, dim=0)
    image_id_per_box = torch.repeat_interleave(
        torch.arange(len(boxes)), torch.tensor([len(b) for b in boxes]))
    images_linear = (images_u8.float() / 255) ** 2.2
    args = (crop_model, mirror_mapping, n_joints, images_linear)
    if boxes_per_batch == 0:
        return predict_single_batch(
            *args, intrinsic_matrix, distortion_coeffs, camspace_up, boxes_flat,
            image_id_per_box, tta, antialias_factor, res)
    n_batches = int(np.ceil(len(boxes_flat) / boxes_per_batch))
    out = []
    for i in range(n_batches):
        s = slice(i * boxes_per_batch, (i + 1) * boxes_per_batch)
        out.append(predict_single_batch(
            *args, intrinsic_matrix[s], distortion_coeffs[s], camspace_up[s], boxes_flat[s],
            image_id_per_box[s], tta, antialias_factor, res))
    return torch.cat(out, dim=0)


def estimate_poses_batched(crop_model, mirror_mapping, n_joints, res, images, boxes,
                           intrinsic_matrix, distortion_coeffs, extrinsic_matrix, world_up_vector,
                           default_fov_degrees=55, internal_batch_size=64, antialias_factor=1,
                           num_aug=5, average_aug=True, skeleton_indices=None,
                           joint_transform_matrix=None):
    """Pose3dEstimator._estimate_poses_batched, multiperson_model.py:76-182.

    images: uint8 [N,3,H,W]; boxes: list of [n_i,5]; camera params as float32 tensors with a
    leading dim of 1 or N.  Returns dict(boxes, poses3d, poses2d) with ragged lists."""
    n_images = len(images)
    if len(intrinsic_matrix) == 1:
        if torch.all(intrinsic_matrix == -1):
            intrinsic_matrix = intrinsic_matrix_from_field_of_view(
                default_fov_degrees, images.shape[2:4])
        intrinsic_matrix = torch.repeat_interleave(intrinsic_matrix, n_images, dim=0)
    if len(distortion_coeffs) == 1:
        distortion_coeffs = torch.repeat_interleave(distortion_coeffs, n_images, dim=0)
    if len(extrinsic_matrix) == 1:
        extrinsic_matrix = torch.repeat_interleave(extrinsic_matrix, n_images, dim=0)

    counts = [len(b) for b in boxes]
    n_box_per_image = torch.tensor(counts)
    intrinsic_matrix = torch.repeat_interleave(intrinsic_matrix, n_box_per_image, dim=0)
    distortion_coeffs = torch.repeat_interleave(distortion_coeffs, n_box_per_image, dim=0)
    camspace_up = torch.einsum('c,bCc->bC', world_up_vector, extrinsic_matrix[..., :3, :3])
    camspace_up = torch.repeat_interleave(camspace_up, n_box_per_image, dim=0)

    tta = tta_params(num_aug)
    poses3d_flat = predict_in_batches(
        crop_model, mirror_mapping, n_joints, images, intrinsic_matrix, distortion_coeffs,
        camspace_up, boxes, internal_batch_size, tta, antialias_factor, res)
    if joint_transform_matrix is not None:
        poses3d_flat = torch.einsum('bank,nN->baNk', poses3d_flat, joint_transform_matrix)

    poses2d_norm = to_homogeneous(distort_points(project(poses3d_flat), distortion_coeffs))
    poses2d_flat = torch.einsum('bank,bjk->banj', poses2d_norm, intrinsic_matrix[:, :2, :])

    inv_ext = torch.repeat_interleave(
        torch.linalg.inv(extrinsic_matrix), n_box_per_image, dim=0)
    poses3d_flat = torch.einsum(
        'bank,bjk->banj', to_homogeneous(poses3d_flat), inv_ext[:, :3, :])
    poses3d = list(torch.split(poses3d_flat, counts))
    poses2d = list(torch.split(poses2d_flat, counts))
    if skeleton_indices is not None:
        poses3d = [p[..., skeleton_indices, :] for p in poses3d]
        poses2d = [p[..., skeleton_indices, :] for p in poses2d]
    if average_aug:
        poses3d = [torch.mean(p, dim=-3) for p in poses3d]
        poses2d = [torch.mean(p, dim=-3) for p in poses2d]
    return dict(boxes=boxes, poses3d=poses3d, poses2d=poses2d)


def mpjpe(a, b):
    """Mean per-joint position error (mean L2 over joints), metrabs_tf/models/eval_metrics.py:17-18
    without root-centering (ours-vs-oracle comparison)."""
    return float(torch.linalg.norm(a.double() - b.double(), dim=-1).mean())


def crop_model_from_features_fp64(features, weight, bias, intrinsics, n_points, cfg,
                                  recombination_weights=None):
    """The same head + reconstruction evaluated in float64 end to end (yardstick only)."""
    if weight.ndim == 2:
        weight = weight[:, :, None, None]
    logits = F.conv2d(features.double(), weight.double(), bias.double())
    c2d, c3d = heads_from_logits(logits, n_points, cfg, eval_dtype=torch.float64)
    if cfg.predict_all_and_latents:
        c2d, c3d = c2d[:, :recombination_weights.shape[0]], c3d[:, :recombination_weights.shape[0]]
    out = reconstruct_absolute(c2d, c3d, intrinsics.double(), cfg)
    if cfg.transform_coords or cfg.predict_all_and_latents:
        out = linear_combine_points(out, recombination_weights.double())
    return out


def postprocess_from_crop_outputs(poses_flat, rot, should_flip, mirror_mapping, intrinsic_matrix,
                                  distortion_coeffs, extrinsic_matrix, joint_transform_matrix=None,
                                  skeleton_indices=None, average_aug=True):
    """Everything the reference does after crop_model for ONE internal batch:
    multiperson_model.py:244-259 (mirror un-swap, @R, transpose) then :143-178 (joint transform,
    2D projection, world transform, skeleton select, TTA mean).  poses_flat [A*n, J, 3];
    rot [A,n,3,3]; camera tensors per box ([n,...]).  Checker for the K7 kernel."""
    num_aug = rot.shape[0]
    poses = torch.reshape(poses_flat, [num_aug, -1, poses_flat.shape[-2], 3])
    swapped = poses[..., mirror_mapping, :]
    poses = torch.where(torch.reshape(should_flip, [-1, 1, 1, 1]), swapped, poses)
    p3 = (poses @ rot).transpose(0, 1)
    if joint_transform_matrix is not None:
        p3 = torch.einsum('bank,nN->baNk', p3, joint_transform_matrix)
    p2n = to_homogeneous(distort_points(project(p3), distortion_coeffs))
    p2 = torch.einsum('bank,bjk->banj', p2n, intrinsic_matrix[:, :2, :])
    p3 = torch.einsum('bank,bjk->banj', to_homogeneous(p3),
                      torch.linalg.inv(extrinsic_matrix)[:, :3, :])
    if skeleton_indices is not None:
        p3, p2 = p3[..., skeleton_indices, :], p2[..., skeleton_indices, :]
    if average_aug:
        p3, p2 = torch.mean(p3, dim=-3), torch.mean(p2, dim=-3)
    return p3, p2


# ------------------------------------------------------------------------------------------------
# Row f.3 -- detector pre-processing and box rescale (the step BEFORE the hot path)
# metrabs_pytorch/multiperson/person_detector.py:14-54.  The detector network itself (ultralytics
# YOLOv8, third party) is out of scope; `detector_preprocess` produces what it is fed and
# `detector_scale_boxes` maps its boxes back to the frame.

def detector_target_size(h, w, input_size=416):
    """person_detector.py:15-20,26-29 -- numpy float32 arithmetic exactly as written there."""
    h32, w32 = np.float32(h), np.float32(w)
    max_side = np.maximum(h32, w32)
    factor = input_size / max_side
    target_w = int(np.int32(factor * w32))
    target_h = int(np.int32(factor * h32))
    pad_h = -target_h % 32
    pad_w = -target_w % 32
    return dict(target_h=target_h, target_w=target_w, antialias=bool(factor < 1),
                pad_top=pad_h // 2, pad_left=pad_w // 2, out_h=target_h + pad_h, out_w=target_w + pad_w,
                x_factor=float(w32 / np.float32(target_w)), y_factor=float(h32 / np.float32(target_h)))


def detector_preprocess(images_u8, input_size=416):
    """person_detector.py:21-33: gamma-decode, bilinear resize (antialiased when shrinking; torchvision's
    tensor path = F.interpolate(bilinear, align_corners=False, antialias)), gamma-encode, pad with 0.5
    to multiples of 32.  images_u8 [N,3,H,W] uint8 -> ([N,3,out_h,out_w] f32, geometry dict)."""
    m = detector_target_size(images_u8.shape[2], images_u8.shape[3], input_size)
    x = (images_u8.float() / 255) ** 2.2
    x = F.interpolate(x, size=[m['target_h'], m['target_w']], mode='bilinear', align_corners=False,
                      antialias=m['antialias'])
    x = x ** (1 / 2.2)
    pad_h, pad_w = m['out_h'] - m['target_h'], m['out_w'] - m['target_w']
    x = F.pad(x, (m['pad_left'], pad_w - m['pad_left'], m['pad_top'], pad_h - m['pad_top']), value=0.5)
    return x, m


def detector_scale_boxes(xyxy_conf, m):
    """person_detector.py:47-54: boxes [n,5] (x1,y1,x2,y2,conf) in the padded network frame ->
    (x, y, w, h, conf) in the original frame."""
    hw, hh = np.float32(m['pad_left']), np.float32(m['pad_top'])
    xf, yf = np.float32(m['x_factor']), np.float32(m['y_factor'])
    b = xyxy_conf
    return torch.stack([(b[:, 0] - hw) * xf, (b[:, 1] - hh) * yf, (b[:, 2] - b[:, 0]) * xf,
                        (b[:, 3] - b[:, 1]) * yf, b[:, 4]], dim=1)


# ------------------------------------------------------------------------------------------------
# Row f.2 -- plausibility filter + pose NMS (the step AFTER the hot path inside detect_poses)
# TF: metrabs_tf/multiperson/multiperson_model.py:441-459 (_filter_poses), plausibility_check.py:9-96.
# PyTorch: metrabs_pytorch/multiperson/plausibility_check.py:8-119 is a port of the same functions,
# but the call site is commented out (multiperson_model.py:158-163,357-378) and
# is_pose_consistent_with_box (:86-103) does not run (torch.min(dim=) returns a tuple).  The
# functions below follow the shared algorithm; where the twins differ the default is noted.

def joint2bone_mat(edges, n_joints):
    """posepile.joint_info.get_joint2bone_mat (third party; published algorithm): one row per
    stick-figure edge, +1 at its first joint and -1 at its second."""
    m = torch.zeros(len(edges), n_joints)
    for b, (j1, j2) in enumerate(edges):
        m[b, int(j1)] = 1
        m[b, int(j2)] = -1
    return m


def is_pose_plausible(poses, edges, mean_bones, n_joints=None):
    """plausibility_check.py:8-29.  poses [...,J,3]; a pose is implausible when some bone is
    (< 0.1x or > 3x its mean length) AND more than 300 mm off."""
    n_joints = n_joints or poses.shape[-2]
    bones = joint2bone_mat(edges, n_joints) @ poses[..., :n_joints, :]
    bone_lengths = torch.norm(bones, dim=-1)
    mean_bones = torch.as_tensor(mean_bones, dtype=torch.float32)
    rel = bone_lengths / mean_bones
    diff = torch.abs(bone_lengths - mean_bones)
    bad = torch.any(((rel > 3) | (rel < 0.1)) & (diff > 300), dim=-1)
    return ~bad


def scale_align(poses):
    """plausibility_check.py:111-114."""
    square_scales = torch.mean(torch.square(poses), dim=(-2, -1), keepdim=True)
    mean_square_scale = torch.mean(square_scales, dim=-3, keepdim=True)
    return poses * torch.sqrt(mean_square_scale / square_scales)


def point_stdev(poses, item_dim, coord_dim, unbiased=False):
    """plausibility_check.py:117-119.  TF's reduce_variance is the population variance
    (unbiased=False, the default here: with num_aug=1 every pose is consistent); the PyTorch port's
    torch.var defaults to the unbiased estimator (NaN at num_aug=1, x sqrt(A/(A-1)) otherwise)."""
    var = torch.var(poses, dim=item_dim, keepdim=True, unbiased=unbiased)
    return torch.squeeze(torch.sqrt(torch.sum(var, dim=coord_dim, keepdim=True)), (item_dim, coord_dim))


def are_augmentation_results_consistent(poses3d, unbiased=False):
    """plausibility_check.py:62-66: at least one fourth of the joints have a stdev under 200 mm."""
    n_joints = poses3d.shape[-2]
    stdevs = point_stdev(scale_align(poses3d), item_dim=1, coord_dim=-1, unbiased=unbiased)
    return torch.count_nonzero(stdevs < 200, dim=1) > (n_joints // 4)


def compute_pose_similarity(poses):
    """plausibility_check.py:69-83 (both twins take the k LARGEST distances)."""
    square_scales = torch.mean(torch.square(poses), dim=(-2, -1), keepdim=True)
    s1, s2 = square_scales.unsqueeze(0), square_scales.unsqueeze(1)
    mean_square_scales = (s1 + s2) / 2
    f1, f2 = torch.sqrt(mean_square_scales / s1), torch.sqrt(mean_square_scales / s2)
    dists = torch.linalg.norm(f1 * poses.unsqueeze(0) - f2 * poses.unsqueeze(1), dim=-1)
    best = torch.topk(dists, k=poses.shape[-2] // 4, sorted=False).values
    return torch.mean(torch.relu(1 - best / 300), dim=-1)


def non_max_suppression_overlaps(overlaps, scores, overlap_threshold, max_output_size=150,
                                 order='index'):
    """Greedy NMS on a similarity matrix, highest score first (stable).  PyTorch port
    (plausibility_check.py:32-52): survivors in ascending index order, no cap (order='index');
    tf.image.non_max_suppression_overlaps (TF :37-39): survivors in descending score order, at most
    max_output_size (order='score')."""
    n = len(overlaps)
    by_score = torch.argsort(scores, stable=True, dim=0, descending=True)
    suppressed = torch.zeros(n, dtype=torch.bool)
    kept = []
    for _i in range(n):
        i = int(by_score[_i])
        if suppressed[i]:
            continue
        if order == 'score' and len(kept) >= max_output_size:
            break
        kept.append(i)
        for _j in range(_i + 1, n):
            j = int(by_score[_j])
            if not suppressed[j] and overlaps[i, j] > overlap_threshold:
                suppressed[j] = True
    kept = torch.tensor(kept, dtype=torch.int64)
    return kept if order == 'score' else torch.sort(kept).values


def pose_non_max_suppression(poses, scores, is_pose_valid, order='index'):
    """plausibility_check.py:55-59."""
    idx = torch.squeeze(torch.argwhere(is_pose_valid), 1)
    sim = compute_pose_similarity(poses[idx])
    keep = non_max_suppression_overlaps(sim, scores[idx], 0.4, order=order)
    return idx[keep]


def is_pose_consistent_with_box(pose2d, box):
    """TF plausibility_check.py:66-84 (the PyTorch port :86-103 raises TypeError as written): the
    intersection of the 2D pose's bounding box with the detection covers more than half of the
    detection."""
    posebox_start = torch.min(pose2d, dim=-2).values
    posebox_end = torch.max(pose2d, dim=-2).values
    box_start, box_end = box[..., :2], box[..., :2] + box[..., 2:4]
    box_area = torch.prod(box[..., 2:4], dim=-1)
    inter = torch.relu(torch.minimum(box_end, posebox_end) - torch.maximum(box_start, posebox_start))
    return torch.prod(inter, dim=-1) > 0.5 * box_area


def filter_poses(boxes, poses3d, poses2d, edges, mean_bones, n_joints=None, unbiased=False,
                 order='index'):
    """TF multiperson_model.py:441-459.  boxes: list of [n_i,5]; poses3d: list of [n_i,A,J,3]
    (camera space, before the skeleton selection); poses2d: list of [n_i,A,J,2].
    -> (list of kept index tensors, list of plausibility masks)."""
    keep, masks = [], []
    for b, p3, p2 in zip(boxes, poses3d, poses2d):
        if len(b) == 0:
            keep.append(torch.zeros(0, dtype=torch.int64))
            masks.append(torch.zeros(0, dtype=torch.bool))
            continue
        p3m, p2m = p3.mean(dim=-3), p2.mean(dim=-3)
        mask = is_pose_plausible(p3m, edges, mean_bones, n_joints) \
            & are_augmentation_results_consistent(p3, unbiased) \
            & is_pose_consistent_with_box(p2m, b)
        masks.append(mask)
        keep.append(pose_non_max_suppression(p3m, b[:, 4], mask, order=order))
    return keep, masks
