"""Explicit configuration of the hot path.

The reference reads a global hydra config at call time (metrabs_pytorch/util.py:41-57; keys from
metrabs_pytorch/config/config.yaml:1-22 and config_s_256.yaml:5-9).  Here the same keys travel as an
explicit object and cross the C-ABI as POD structs (include/metrabs_hip.h), so nothing global is
read inside a kernel launch.
"""
import dataclasses
from typing import Any, Optional

from metrabs_amd import _lib


@dataclasses.dataclass
class MetrabsConfig:
    proc_side: int = 256
    stride_train: int = 32
    stride_test: int = 32
    centered_stride: bool = True
    legacy_centered_stride_bug: bool = False
    depth: int = 8
    box_size_mm: float = 2200.0
    weak_perspective: bool = False
    mix_3d_inside_fov: Optional[float] = 0.5
    # the affine-latent options of the crop model (config.yaml; models/metrabs.py:23-44,52-62)
    affine_weights: Any = None          # path / name of the .npz with w1, w2 -- or a dict holding them
    transform_coords: bool = False
    predict_all_and_latents: bool = False
    regularize_to_manifold: bool = False

    def head_params(self):
        return _lib.HeadParams(
            proc_side=self.proc_side, stride_test=self.stride_test,
            centered_stride=int(self.centered_stride),
            legacy_centered_stride_bug=int(self.legacy_centered_stride_bug),
            box_size_mm=float(self.box_size_mm))

    def recon_params(self, mix_3d_inside_fov='cfg', weak_perspective=None):
        mix = self.mix_3d_inside_fov if mix_3d_inside_fov == 'cfg' else mix_3d_inside_fov
        weak = self.weak_perspective if weak_perspective is None else weak_perspective
        return _lib.ReconParams(
            proc_side=self.proc_side, stride_train=self.stride_train,
            centered_stride=int(self.centered_stride), weak_perspective=int(bool(weak)),
            mix_enabled=int(mix is not None), mix_3d_inside_fov=float(mix or 0.0),
            l2_reg=1e-2, weight_eps=1e-4, fov_border_factor=0.75)

    @classmethod
    def from_any(cls, obj):
        """Accepts a MetrabsConfig, a dict, or any object with the reference's config keys
        (e.g. the hydra/OmegaConf object the reference passes around)."""
        if isinstance(obj, cls):
            return obj
        names = [f.name for f in dataclasses.fields(cls)]
        if isinstance(obj, dict):
            return cls(**{k: obj[k] for k in names if k in obj})
        return cls(**{k: getattr(obj, k) for k in names if hasattr(obj, k)})


# Shipped configurations of the reference (metrabs_pytorch/config/*.yaml)
CONFIG_S_256 = MetrabsConfig(proc_side=256, centered_stride=False, legacy_centered_stride_bug=True)
CONFIG_L_384 = MetrabsConfig(proc_side=384)
CONFIG_DEFAULT = MetrabsConfig()
