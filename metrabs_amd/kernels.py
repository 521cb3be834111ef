"""Tensor-level wrappers of the C-ABI (include/metrabs_hip.h): torch tensors in, torch tensors out,
kernels enqueued on torch's current HIP stream.  torch is used for device memory and streams only.
"""
import ctypes

import torch

from metrabs_amd import _lib
from metrabs_amd._lib import check, current_stream_ptr, dtype_code, require_cuda


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def softargmax_decode(logits, n_points, cfg, out=None):
    """logits [B, J*(1+D), H, W] (f32/f16/bf16, NCHW) -> (coords2d [B,J,2] px, coords3d_rel [B,J,3] mm).
    MetrabsHeads.forward after the conv (metrabs_pytorch/models/metrabs.py:78-85)."""
    require_cuda(logits)
    lib = _lib.load()
    logits = logits.contiguous()
    B, n_out, H, W = logits.shape
    J = int(n_points)
    D = n_out // J - 1
    if n_out != J * (1 + D) or D != cfg.depth:
        raise ValueError(f'logits have {n_out} channels, expected J*(1+depth) = {J * (1 + cfg.depth)}')
    if out is None:
        c2d = torch.empty(B, J, 2, device=logits.device, dtype=torch.float32)
        c3d = torch.empty(B, J, 3, device=logits.device, dtype=torch.float32)
    else:
        c2d, c3d = out
    hp = cfg.head_params()
    check(lib.mtr_softargmax_decode(
        _ptr(logits), dtype_code(logits.dtype), _lib.MTR_NCHW, B, J, D, H, W, ctypes.byref(hp),
        _ptr(c2d), _ptr(c3d), current_stream_ptr(logits.device)), 'mtr_softargmax_decode')
    return c2d, c3d


def reconstruct_workspace(B, J, device):
    n = _lib.load().mtr_reconstruct_workspace_bytes(int(B), int(J))
    return torch.empty((n + 7) // 8, device=device, dtype=torch.float64)


def reconstruct_absolute(coords2d, coords3d_rel, intrinsics, cfg, mix_3d_inside_fov='cfg',
                         weak_perspective=None, workspace=None, out=None):
    """ptu3d.reconstruct_absolute (metrabs_pytorch/ptu3d.py:9-33)."""
    require_cuda(coords2d, coords3d_rel, intrinsics)
    lib = _lib.load()
    coords2d = coords2d.contiguous().float()
    coords3d_rel = coords3d_rel.contiguous().float()
    intrinsics = intrinsics.contiguous().float()
    B, J = coords2d.shape[:2]
    if intrinsics.shape != (B, 3, 3) or coords3d_rel.shape != (B, J, 3):
        raise ValueError('shape mismatch between coords2d, coords3d_rel and intrinsics')
    if workspace is None:
        workspace = reconstruct_workspace(B, J, coords2d.device)
    if out is None:
        out = torch.empty(B, J, 3, device=coords2d.device, dtype=torch.float32)
    rp = cfg.recon_params(mix_3d_inside_fov, weak_perspective)
    check(lib.mtr_reconstruct_absolute(
        _ptr(coords2d), _ptr(coords3d_rel), _ptr(intrinsics), B, J, ctypes.byref(rp), _ptr(out),
        _ptr(workspace), workspace.numel() * 8, current_stream_ptr(coords2d.device)),
        'mtr_reconstruct_absolute')
    return out


def reconstruct_moments(coords2d, coords3d_rel, intrinsics, workspace=None):
    """-> f64 [3] = (sum normalized2d^2, sum rel_backproj^2, count) over this call's crops."""
    require_cuda(coords2d, coords3d_rel, intrinsics)
    lib = _lib.load()
    B, J = coords2d.shape[:2]
    if workspace is None:
        workspace = reconstruct_workspace(B, J, coords2d.device)
    moments = torch.zeros(3, device=coords2d.device, dtype=torch.float64)
    if B > 0:
        check(lib.mtr_reconstruct_moments(
            _ptr(coords2d), _ptr(coords3d_rel), _ptr(intrinsics), B, J, _ptr(moments),
            _ptr(workspace), workspace.numel() * 8, current_stream_ptr(coords2d.device)),
            'mtr_reconstruct_moments')
    return moments


def reconstruct_solve(coords2d, coords3d_rel, intrinsics, moments, cfg, mix_3d_inside_fov='cfg',
                      weak_perspective=None, out=None):
    require_cuda(coords2d, coords3d_rel, intrinsics, moments)
    lib = _lib.load()
    B, J = coords2d.shape[:2]
    if out is None:
        out = torch.empty(B, J, 3, device=coords2d.device, dtype=torch.float32)
    rp = cfg.recon_params(mix_3d_inside_fov, weak_perspective)
    check(lib.mtr_reconstruct_solve(
        _ptr(coords2d), _ptr(coords3d_rel), _ptr(intrinsics), B, J, ctypes.byref(rp),
        _ptr(moments), _ptr(out), current_stream_ptr(coords2d.device)), 'mtr_reconstruct_solve')
    return out
