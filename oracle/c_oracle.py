"""TEST INFRASTRUCTURE: ctypes access to the plain-C oracle (oracle/mtr_oracle.c)."""
import ctypes

import numpy as np

from oracle import build_c


class OrcConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        'proc_side', 'stride_train', 'stride_test', 'centered_stride', 'legacy_centered_stride_bug',
        'weak_perspective', 'mix_enabled')] + [('box_size_mm', ctypes.c_float),
                                               ('mix_3d_inside_fov', ctypes.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build_c.build())
    return _lib


def _cfg(cfg):
    mix = cfg.mix_3d_inside_fov
    return OrcConfig(cfg.proc_side, cfg.stride_train, cfg.stride_test, int(cfg.centered_stride),
                     int(cfg.legacy_centered_stride_bug), int(cfg.weak_perspective),
                     int(mix is not None), float(cfg.box_size_mm), float(mix or 0.0))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def decode(logits, n_points, cfg):
    logits = _f32(logits)
    B, n_out, H, W = logits.shape
    D = n_out // n_points - 1
    c2d = np.empty((B, n_points, 2), np.float32)
    c3d = np.empty((B, n_points, 3), np.float32)
    c = _cfg(cfg)
    lib().orc_decode(_p(logits), B, n_points, D, H, W, ctypes.byref(c), _p(c2d), _p(c3d))
    return c2d, c3d


def reconstruct(coords2d, coords3d_rel, K, cfg):
    c2d, rel, K = _f32(coords2d), _f32(coords3d_rel), _f32(K)
    B, J = c2d.shape[:2]
    out = np.empty((B, J, 3), np.float32)
    c = _cfg(cfg)
    rc = lib().orc_reconstruct(_p(c2d), _p(rel), _p(K), B, J, ctypes.byref(c), _p(out))
    if rc != 0:
        raise RuntimeError(f'orc_reconstruct returned {rc}')
    return out


def pyramid(images_u8):
    img = np.ascontiguousarray(images_u8, dtype=np.uint8)
    n, c, h, w = img.shape
    l0 = np.empty((n, c, h, w), np.float32)
    l1 = np.empty((n, c, h // 2, w // 2), np.float32)
    l2 = np.empty((n, c, h // 4, w // 4), np.float32)
    lib().orc_pyramid(_p(img), n * c, h, w, _p(l0), _p(l1), _p(l2))
    return l0, l1, l2


def warp(level_image, k_level, hinv, dist, res):
    img = _f32(level_image)
    d12 = np.zeros(12, np.float32)
    d12[:len(dist)] = dist
    out = np.empty((3, res, res), np.float32)
    lib().orc_warp(_p(img), img.shape[1], img.shape[2], _p(_f32(k_level)), _p(_f32(hinv)), _p(d12),
                   res, _p(out))
    return out
