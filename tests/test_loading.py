"""Row f.4: the model-directory format and the checkpoint key layout.

* the EfficientNetV2 restatement has the reference's parameter names / shapes and arithmetic:
  golden minted from the reference's own class (tests/golden/backbone_effnetv2_*.npz) and, where
  /root/reference is mounted, the live class;
* a model directory written with the reference's five files loads back bit-identically;
* TF -> PyTorch tensor layout mapping (convert_model_from_tf.py:89-98)."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import load_golden, same_cpu_as_golden
from oracle import cases
from oracle import ref_harness as rh


@pytest.mark.parametrize('size', ['s', 'l'])
def test_backbone_keys_shapes_and_output_vs_golden(size):
    from metrabs_amd import backbones
    g = load_golden(f'backbone_effnetv2_{size}')
    net = backbones.efficientnetv2(size).eval()
    sd = net.state_dict()
    assert list(sd) == [str(k) for k in g['keys']]
    assert [','.join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g['shapes']]
    net.load_state_dict(cases.deterministic_state(sd))
    with torch.inference_mode():
        y = net(cases.backbone_probe_input()).numpy()
    np.testing.assert_allclose(y, g['output'], rtol=2e-4 if not same_cpu_as_golden(g) else 1e-5,
                               atol=2e-5 if not same_cpu_as_golden(g) else 1e-6)


@pytest.mark.skipif(not rh.reference_available(), reason='/root/reference not mounted')
def test_backbone_vs_live_reference_class():
    from metrabs_amd import backbones
    ref = rh.load()
    with rh.config():
        theirs = torch.nn.Sequential(ref.efficientnet.PreprocLayer(),
                                     ref.efficientnet.efficientnet_v2_s().features).eval()
    ours = backbones.efficientnetv2('s').eval()
    state = cases.deterministic_state(theirs.state_dict(), seed=3)
    theirs.load_state_dict(state)
    ours.load_state_dict(state)  # strict: same keys
    x = torch.rand(1, 3, 128, 160, generator=cases.gen(4))
    # (symmetric padding is folded into the convolution here and explicit in the reference: oneDNN
    #  may pick different blockings for the two, hence a rounding-level tolerance, not equality)
    with torch.inference_mode():
        torch.testing.assert_close(ours(x), theirs(x), rtol=1e-5, atol=1e-6)
    # centered_stride = False: no bottom-right shift on the last stride-2 stage
    with rh.config(centered_stride=False):
        theirs = torch.nn.Sequential(ref.efficientnet.PreprocLayer(),
                                     ref.efficientnet.efficientnet_v2_s().features).eval()
    ours = backbones.efficientnetv2('s', centered_stride=False).eval()
    theirs.load_state_dict(state)
    ours.load_state_dict(state)
    with torch.inference_mode():
        torch.testing.assert_close(ours(x), theirs(x), rtol=1e-5, atol=1e-6)


def test_model_directory_round_trip(tmp_path):
    """demo_image.py:49-74: config.yaml, ckpt.pt, joint_info.npz, skeleton_infos.pkl,
    joint_transform_matrix.npy -> Metrabs with the same parameters."""
    from metrabs_amd import backbones, loading
    from metrabs_amd.config import MetrabsConfig
    from metrabs_amd.joint_info import JointInfo
    from metrabs_amd.models.metrabs import Metrabs
    raw = dict(proc_side=256, stride_train=32, stride_test=32, centered_stride=True, depth=8,
               box_size_mm=2200, backbone='efficientnetv2-s', efficientnet_size='s',
               weak_perspective=False, mix_3d_inside_fov=0.5, load_path=None)
    bb = backbones.efficientnetv2('s')
    model = Metrabs(bb, JointInfo(cases.COCO17, cases.COCO17_EDGES), MetrabsConfig.from_any(raw),
                    in_channels=bb.out_channels)
    model.load_state_dict(cases.deterministic_state(model.state_dict(), seed=9))
    skel = {'': dict(indices=list(range(17)), names=cases.COCO17, edges=cases.COCO17_EDGES),
            'upper': dict(indices=[0, 5, 6, 7, 8], names=['nose', 'lsho', 'rsho', 'lelb', 'relb'],
                          edges=[[1, 3], [2, 4]])}
    jtm = np.eye(17, dtype=np.float32)
    d = str(tmp_path / 'model')
    loading.save_model_dir(d, model, raw, skel, jtm)
    assert sorted(os.listdir(d)) == ['ckpt.pt', 'config.yaml', 'joint_info.npz',
                                     'joint_transform_matrix.npy', 'skeleton_infos.pkl']
    state = torch.load(os.path.join(d, 'ckpt.pt'))
    assert 'backbone.1.0.0.weight' in state and 'heatmap_heads.conv_final.weight' in state
    assert state['heatmap_heads.conv_final.weight'].shape == (17 * 9, 1280, 1, 1)
    loaded = loading.load_crop_model(d)
    assert not loaded.training
    assert loaded.joint_info.names == cases.COCO17
    for (k1, v1), (k2, v2) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    cfg, raw2 = loading.load_config(d)
    assert cfg.proc_side == 256 and cfg.depth == 8 and raw2['efficientnet_size'] == 's'
    # the inference copy with folded batch norms: no BatchNorm2d left, same head parameters
    folded = loading.load_crop_model(d, fold_batchnorm=True)
    assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in folded.backbone.modules())
    assert torch.equal(folded.heatmap_heads.conv_final.weight, loaded.heatmap_heads.conv_final.weight)
    with open(os.path.join(d, 'skeleton_infos.pkl'), 'rb') as f:
        assert pickle.load(f)['upper']['indices'] == [0, 5, 6, 7, 8]
    # a checkpoint with a missing or renamed key must not load silently
    del state['backbone.1.3.0.block.1.0.weight']
    torch.save(state, os.path.join(d, 'ckpt.pt'))
    with pytest.raises(RuntimeError):
        loading.load_crop_model(d)


def test_model_directory_with_affine_weights(tmp_path):
    """Row a11 through the loader: config.yaml names the affine-weights file (models/metrabs.py:23-32); a file of
    that name inside the model directory is found, the head gets the mode's raw point count and the checkpoint
    loads strictly."""
    from metrabs_amd import backbones, loading
    from metrabs_amd.config import MetrabsConfig
    from metrabs_amd.joint_info import JointInfo
    from metrabs_amd.models.metrabs import Metrabs
    w1, w2 = cases.affine_weights_case(17, 12, 77)
    raw = dict(proc_side=256, depth=8, backbone='resnet18', affine_weights='latents12', predict_all_and_latents=True)
    bb = backbones.build_backbone('resnet18')
    model = Metrabs(bb, JointInfo(cases.COCO17, cases.COCO17_EDGES), MetrabsConfig.from_any(raw),
                    in_channels=bb.out_channels, affine_weights=dict(w1=w1, w2=w2))
    assert model.heatmap_heads.conv_final.out_channels == (12 + 17) * 9
    d = str(tmp_path / 'model')
    loading.save_model_dir(d, model, raw, {'': dict(indices=list(range(17)), names=cases.COCO17,
                                                    edges=cases.COCO17_EDGES)}, np.eye(17, dtype=np.float32))
    with pytest.raises(FileNotFoundError):      # the named file is nowhere: the reference fails the same way
        loading.load_crop_model(d)
    np.savez(os.path.join(d, 'latents12.npz'), w1=w1.numpy(), w2=w2.numpy())
    loaded = loading.load_crop_model(d)
    assert loaded.n_latents == 12 and loaded.latent_prefix == 12 and loaded.latent_output
    assert torch.equal(loaded.recombination_weights, w2)
    assert loaded.heatmap_heads.conv_final.weight.shape[0] == 29 * 9
    for (k1, v1), (k2, v2) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_tf_to_pt_layouts():
    from metrabs_amd import loading
    g = np.random.default_rng(0)
    k = g.standard_normal((3, 3, 8, 16)).astype(np.float32)   # h w c_in c_out
    assert loading.rearrange_tf_to_pt(k).shape == (16, 8, 3, 3)
    assert loading.rearrange_tf_to_pt(k)[5, 2, 1, 0] == k[1, 0, 2, 5]
    dw = g.standard_normal((3, 3, 8, 1)).astype(np.float32)   # h w c mult
    assert loading.rearrange_tf_to_pt(dw, depthwise=True).shape == (8, 1, 3, 3)
    assert loading.rearrange_tf_to_pt(dw, depthwise=True)[4, 0, 2, 1] == dw[2, 1, 4, 0]
    dense = g.standard_normal((8, 16)).astype(np.float32)
    assert np.array_equal(loading.rearrange_tf_to_pt(dense), dense.T)
    assert loading.rearrange_tf_to_pt(np.arange(4.0)).shape == (4,)
    head = loading.head_weights_from_tf(g.standard_normal((1, 1, 1280, 153)).astype(np.float32),
                                        np.zeros(153, np.float32))
    assert head['heatmap_heads.conv_final.weight'].shape == (153, 1280, 1, 1)


@pytest.mark.gpu
def test_loaded_model_directory_runs_the_hot_path(tmp_path, hip_lib):
    """load_multiperson_model (demo_image.py:49-56) -> estimate_poses on the GPU, and the head of the
    loaded model equals the oracle on the loaded weights."""
    from metrabs_amd import backbones, loading
    from metrabs_amd.config import MetrabsConfig
    from metrabs_amd.joint_info import JointInfo
    from metrabs_amd.models.metrabs import Metrabs
    from oracle import cpu_ref
    raw = dict(proc_side=256, stride_train=32, stride_test=32, centered_stride=True, depth=8,
               box_size_mm=2200, efficientnet_size='s', weak_perspective=False, mix_3d_inside_fov=0.5)
    bb = backbones.efficientnetv2('s')
    model = Metrabs(bb, JointInfo(cases.COCO17, cases.COCO17_EDGES), MetrabsConfig.from_any(raw),
                    in_channels=bb.out_channels)
    model.load_state_dict(cases.deterministic_state(model.state_dict(), seed=11))
    skel = {'': dict(indices=list(range(17)), names=cases.COCO17, edges=cases.COCO17_EDGES)}
    d = str(tmp_path / 'model')
    loading.save_model_dir(d, model, raw, skel, np.eye(17, dtype=np.float32))
    est = loading.load_multiperson_model(d)
    img = cases.synth_images(1, 240, 320, 5)[0]
    with torch.inference_mode():
        pred = est.estimate_poses(img.cuda(), torch.tensor([[60.0, 20.0, 120.0, 200.0]]), num_aug=2)
        feats = est.crop_model.backbone(torch.rand(2, 3, 256, 256, device='cuda', generator=None))
        c2d, c3d = est.crop_model.heatmap_heads(feats)
        w = est.crop_model.heatmap_heads.conv_final.weight.detach().cpu()[:, :, 0, 0]
        b = est.crop_model.heatmap_heads.conv_final.bias.detach().cpu()
        o2d, o3d = cpu_ref.heads_forward(feats.cpu(), w, b, 17, cpu_ref.HeadConfig())
    assert pred['poses3d'].shape == (1, 17, 3) and bool(torch.isfinite(pred['poses3d']).all())
    assert cpu_ref.mpjpe(c3d.cpu(), o3d) <= 1e-3
