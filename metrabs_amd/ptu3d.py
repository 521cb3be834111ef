"""Drop-in for metrabs_pytorch/ptu3d.py:reconstruct_absolute (the only function of that module on the
hot path): same name and argument meaning, the work happens in metrabs_amd/csrc/reconstruct.hip."""
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
