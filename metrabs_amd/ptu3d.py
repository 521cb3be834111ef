"""Drop-in for the hot-path part of metrabs_pytorch/ptu3d.py.  reconstruct_absolute runs in the HIP
kernels of metrabs_amd/csrc/reconstruct.hip; the O(1)-sized camera helpers stay as torch ops."""
import numpy as np
import torch

from metrabs_amd import kernels
from metrabs_amd.config import CONFIG_DEFAULT, MetrabsConfig


def reconstruct_absolute(coords2d, coords3d_rel, intrinsics, mix_3d_inside_fov=None,
                         weak_perspective=None, config=None):
    """ptu3d.reconstruct_absolute (ptu3d.py:9-33).  As in the reference, mix_3d_inside_fov=None
    means "no mixing"; Metrabs.forward passes config.mix_3d_inside_fov (models/metrabs.py:57-59)."""
    cfg = MetrabsConfig.from_any(config) if config is not None else CONFIG_DEFAULT
    return kernels.reconstruct_absolute(
        coords2d, coords3d_rel, intrinsics, cfg, mix_3d_inside_fov=mix_3d_inside_fov,
        weak_perspective=weak_perspective)


def to_homogeneous(x):
    """ptu3d.py:52-53."""
    return torch.cat([x, torch.ones_like(x[..., :1])], dim=-1)


def project(points):
    """ptu3d.py:145-146."""
    return points[..., :2] / points[..., 2:3]


def intrinsic_matrix_from_field_of_view(fov_degrees, imshape):
    """ptu3d.py:149-161."""
    imshape = torch.tensor(imshape, dtype=torch.float32)
    fov_radians = fov_degrees * torch.tensor(np.pi / 180, dtype=torch.float32)
    focal_length = torch.max(imshape) / (torch.tan(fov_radians / 2) * 2)
    zero, one = torch.tensor(0, dtype=torch.float32), torch.tensor(1, dtype=torch.float32)
    return torch.stack([
        torch.stack([focal_length, zero, imshape[1] / 2], dim=-1),
        torch.stack([zero, focal_length, imshape[0] / 2], dim=-1),
        torch.stack([zero, zero, one], dim=-1)], dim=-2).unsqueeze(0)


def rotation_mat(angle, rot_axis):
    """ptu3d.py:164-184."""
    sin, cos = torch.sin(angle), torch.cos(angle)
    zero, one = torch.zeros_like(angle), torch.ones_like(angle)
    rows = {
        'x': [[one, zero, zero], [zero, cos, sin], [zero, -sin, cos]],
        'y': [[cos, zero, -sin], [zero, one, zero], [sin, zero, cos]],
        'z': [[cos, -sin, zero], [sin, cos, zero], [zero, zero, one]]}[rot_axis]
    return torch.stack([torch.stack(r, dim=-1) for r in rows], dim=-2)
