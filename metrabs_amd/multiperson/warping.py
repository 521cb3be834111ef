"""Drop-in for metrabs_pytorch/multiperson/warping.py: same function names and argument meaning,
backed by the HIP sampler (metrabs_amd/csrc/warp.hip) -- one launch per call instead of a Python
loop with ~10 launches per crop (warping.py:23-28)."""
import numpy as np
import torch

from metrabs_amd import _lib, kernels


def pad_axis_to_size(x, size, axis=-1):
    """warping.pad_axis_to_size (warping.py:110-113) for the last axis."""
    assert axis in (-1, x.ndim - 1)
    return torch.nn.functional.pad(x, (0, size - x.shape[-1]))


def corner_aligned_scale_mat(factor):
    """warping.corner_aligned_scale_mat (warping.py:128-133)."""
    shift = (factor - 1) / 2
    return torch.from_numpy(np.array(
        [[factor, 0, shift], [0, factor, shift], [0, 0, 1]], dtype=np.float32))


def make_warp_params(intrinsic_matrix, new_invprojmats, distortion_coeffs, crop_scales, image_ids,
                     gamma_exponents=None, n_pyramid_levels=3):
    """Packs the per-crop rows mtr_warp_crops consumes (include/metrabs_hip.h) from the arguments of
    warp_images_with_pyramid: level = clip(floor(-log2(scale)), 0, n-1) (warping.py:20-21),
    K_level = corner_aligned_scale_mat(2^-l) @ K (warping.py:15-17)."""
    dev = new_invprojmats.device
    n = new_invprojmats.shape[0]
    levels = torch.clip(torch.floor(-torch.log2(crop_scales.float())), 0, n_pyramid_levels - 1)
    scale_mats = torch.stack([corner_aligned_scale_mat(1 / 2 ** l) for l in range(n_pyramid_levels)])
    k_lvl = scale_mats.to(dev)[levels.long()] @ intrinsic_matrix.float()
    d12 = pad_axis_to_size(distortion_coeffs.float(), 12)
    wp = torch.zeros(n, _lib.MTR_WARP_PARAM_FLOATS, device=dev, dtype=torch.float32)
    wp[:, 0:9] = new_invprojmats.reshape(n, 9)
    wp[:, 9:18] = k_lvl.reshape(n, 9)
    wp[:, 18:30] = d12
    wp[:, 30] = (d12 != 0).any(dim=1).float()
    wp[:, 31] = levels
    wp[:, 32] = image_ids.float()
    wp[:, 33] = 1.0 if gamma_exponents is None else gamma_exponents
    wp[:, 34] = crop_scales
    return wp


def warp_images_with_pyramid(images, intrinsic_matrix, new_invprojmats, distortion_coeffs,
                             crop_scales, output_shape, image_ids, n_pyramid_levels=3):
    """warping.warp_images_with_pyramid (warping.py:6-28).

    images: f32 linear-light [N,3,H,W] on the GPU, or an already built kernels.Pyramid.
    n_pyramid_levels 1..3: the level of a crop is clipped to n - 1 (warping.py:20-21), so fewer levels are the
    3-level pyramid with the coarse ones never chosen.  output_shape (h, w): the sampler produces squares; a
    rectangle is the top-left h x w of the max(h, w) square (an output pixel's source position does not depend
    on the output size, warping.py:41-54) -- the reference's own callers only ask for res x res."""
    n_pyramid_levels = int(n_pyramid_levels)
    if not 1 <= n_pyramid_levels <= 3:
        raise NotImplementedError('the HIP sampler holds a pyramid of 3 levels (the reference default); '
                                  f'n_pyramid_levels={n_pyramid_levels}')
    out_h, out_w = int(output_shape[0]), int(output_shape[1])
    pyr = images if isinstance(images, kernels.Pyramid) else kernels.pyramid_from_level0(images)
    dev = pyr.device
    wp = make_warp_params(
        intrinsic_matrix.to(dev), new_invprojmats.to(dev), distortion_coeffs.to(dev),
        crop_scales.to(dev), image_ids.to(dev), None, n_pyramid_levels)
    crops = kernels.warp_crops(pyr, wp, max(out_h, out_w), antialias=1)
    return crops if out_h == out_w else crops[..., :out_h, :out_w].contiguous()
