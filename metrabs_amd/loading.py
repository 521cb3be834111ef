"""Model-directory loader -- the on-disk format of the released models.

Mirrors metrabs_pytorch/scripts/demo_image.py:49-74:

    <model_dir>/config.yaml                  the hydra config the model was exported with
    <model_dir>/ckpt.pt                      state_dict of Metrabs: 'backbone.1.<...>' (EfficientNetV2
                                             features behind the PreprocLayer at index 0) and
                                             'heatmap_heads.conv_final.{weight,bias}'
    <model_dir>/joint_info.npz               joint_names, joint_edges
    <model_dir>/skeleton_infos.pkl           {skeleton name: {indices, names, edges}}
    <model_dir>/joint_transform_matrix.npy   [n_model_joints, n_output_joints]

plus the TF -> PyTorch tensor layout mapping of convert_model_from_tf.py:89-98 for weights that come
straight from a TF checkpoint.  The backbone restatement (metrabs_amd/backbones.py) has the
reference's module tree, so `ckpt.pt` loads with strict key matching.
"""
import os
import pickle

import numpy as np
import torch
import yaml

from metrabs_amd import backbones
from metrabs_amd.config import MetrabsConfig
from metrabs_amd.joint_info import JointInfo
from metrabs_amd.models.metrabs import Metrabs
from metrabs_amd.multiperson.multiperson_model import Pose3dEstimator


def load_config(model_dir):
    """config.yaml -> (MetrabsConfig, raw dict).  The reference composes it with hydra
    (util.py:40-60); the exported file is a flat mapping, read here with a plain YAML parser."""
    with open(os.path.join(model_dir, 'config.yaml')) as f:
        raw = yaml.safe_load(f) or {}
    return MetrabsConfig.from_any(raw), raw


def backbone_from_config(raw):
    """demo_image.py:63-66: `efficientnet_v2_<cfg.efficientnet_size>` behind a PreprocLayer; the
    `backbone:` key (config.yaml:10, e.g. 'efficientnetv2-s') is accepted as well."""
    centered = bool(raw.get('centered_stride', True))
    size = raw.get('efficientnet_size')
    if size is None:
        name = str(raw.get('backbone', 'efficientnetv2-s')).lower()
        if not name.startswith('efficientnetv2'):
            return backbones.build_backbone(name)
        size = name.rsplit('-', 1)[-1]
    return backbones.efficientnetv2(str(size), centered_stride=centered)


def load_joint_info(model_dir):
    ji = np.load(os.path.join(model_dir, 'joint_info.npz'))
    return JointInfo(ji['joint_names'], ji['joint_edges'])


def load_crop_model(model_dir, map_location='cpu', fold_batchnorm=False, fused_epilogue=False):
    """demo_image.py:59-74 -> Metrabs in eval mode with the checkpoint loaded (strict).
    fold_batchnorm=True then replaces the backbone by its inference copy with every batch norm
    folded into the convolution in front of it (backbones.fold_batchnorm: the same function up to
    rounding, ~12 % less backbone time; fused_epilogue=True also runs bias + activation behind the
    folded convolutions as one in-place HIP pass, K10); the default keeps the checkpoint's own
    arithmetic."""
    cfg, raw = load_config(model_dir)
    # config.affine_weights (models/metrabs.py:23-32) is a path or a name under $DATA_ROOT/skeleton_conversion;
    # a file of that name shipped INSIDE the model directory is found too
    if isinstance(cfg.affine_weights, str) and cfg.affine_weights and not os.path.exists(cfg.affine_weights):
        for cand in (cfg.affine_weights, cfg.affine_weights + '.npz', 'affine_weights.npz'):
            if os.path.exists(os.path.join(model_dir, cand)):
                cfg.affine_weights = os.path.join(model_dir, cand)
                break
    backbone = backbone_from_config(raw)
    # (the reference materialises its LazyConv2d head with a dummy forward, demo_image.py:69-72;
    #  the channel count is known here)
    model = Metrabs(backbone, load_joint_info(model_dir), cfg, in_channels=backbone.out_channels)
    state = torch.load(os.path.join(model_dir, 'ckpt.pt'), map_location=map_location)
    model.load_state_dict(state, strict=True)
    model = model.eval()
    if fold_batchnorm:
        from .backbones import fold_batchnorm as fold
        model.backbone = fold(model.backbone, fused_epilogue=fused_epilogue)
    return model


def load_multiperson_model(model_dir, device='cuda', detector=None, fold_batchnorm=False,
                           fused_epilogue=False):
    """demo_image.py:49-56 -> Pose3dEstimator on `device`."""
    model = load_crop_model(model_dir, fold_batchnorm=fold_batchnorm, fused_epilogue=fused_epilogue)
    with open(os.path.join(model_dir, 'skeleton_infos.pkl'), 'rb') as f:
        skeleton_infos = pickle.load(f)
    joint_transform_matrix = np.load(os.path.join(model_dir, 'joint_transform_matrix.npy'))
    return Pose3dEstimator(model.to(device), skeleton_infos, joint_transform_matrix, detector=detector)


def save_model_dir(model_dir, model, config_dict, skeleton_infos, joint_transform_matrix):
    """Writes the same five files (used by the tests and for re-exporting converted weights)."""
    os.makedirs(model_dir, exist_ok=True)
    with open(os.path.join(model_dir, 'config.yaml'), 'w') as f:
        yaml.safe_dump(dict(config_dict), f)
    torch.save(model.state_dict(), os.path.join(model_dir, 'ckpt.pt'))
    ji = model.joint_info
    np.savez(os.path.join(model_dir, 'joint_info.npz'), joint_names=np.array(ji.names),
             joint_edges=np.array(ji.stick_figure_edges))
    with open(os.path.join(model_dir, 'skeleton_infos.pkl'), 'wb') as f:
        pickle.dump(skeleton_infos, f)
    np.save(os.path.join(model_dir, 'joint_transform_matrix.npy'), np.asarray(joint_transform_matrix))


def rearrange_tf_to_pt(value, depthwise=False):
    """convert_model_from_tf.py:89-98: TF kernels are [h, w, c_in, c_out] (depthwise:
    [h, w, c, multiplier]), dense kernels [c_in, c_out]; PyTorch wants [c_out, c_in, h, w] /
    [c, multiplier, h, w] / [c_out, c_in]."""
    value = np.asarray(value)
    if value.ndim == 4:
        return value.transpose(2, 3, 0, 1) if depthwise else value.transpose(3, 2, 0, 1)
    if value.ndim == 2:
        return value.transpose(1, 0)
    return value


def head_weights_from_tf(kernel, bias):
    """The heatmap head of a TF checkpoint ('metrabs/metrabs_heads/conv2d/{kernel,bias}:0',
    convert_model_from_tf.py:175-177,194) -> entries of the PyTorch state_dict.  The channel order
    (2D maps first, then depth-major 3D slices) is the same in both frameworks
    (tf models/metrabs.py:100-101, pt models/metrabs.py:78-79)."""
    return {'heatmap_heads.conv_final.weight': torch.from_numpy(
                np.ascontiguousarray(rearrange_tf_to_pt(kernel))),
            'heatmap_heads.conv_final.bias': torch.from_numpy(np.asarray(bias))}
