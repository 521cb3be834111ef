"""CPU: host-side mirror of the reference interface (no GPU, no compute kernels)."""
import inspect

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import cases, cpu_ref


def test_tta_parameters_match_reference_golden():
    from metrabs_amd.multiperson.multiperson_model import tta_parameters
    g = load_golden('tta_params')
    for num_aug in range(1, 7):
        t = tta_parameters(num_aug)
        for k in ('gammas', 'scales', 'should_flip', 'rotflipmat'):
            assert np.array_equal(t[k].numpy(), g[f'a{num_aug}_{k}']), (num_aug, k)


def test_linspace_semantics():
    from metrabs_amd.multiperson.multiperson_model import tta_linspace
    assert tta_linspace(0.6, 1.0, 1).tolist() == pytest.approx([0.8])       # midpoint
    assert tta_linspace(0.8, 1.0, 2, endpoint=False).tolist() == pytest.approx([0.8, 0.9])
    for n in (1, 2, 5):
        for ep in (True, False):
            assert torch.equal(tta_linspace(0.8, 1.0, n, endpoint=ep),
                               cpu_ref.ref_linspace(0.8, 1.0, n, endpoint=ep))


def test_joint_info_mirror_mapping():
    from metrabs_amd.joint_info import JointInfo
    ji = JointInfo(cases.COCO17, cases.COCO17_EDGES)
    assert ji.n_joints == 17
    assert np.array_equal(ji.mirror_mapping, cases.mirror_mapping(cases.COCO17))
    assert ji.mirror_mapping[cases.COCO17.index('lwri')] == cases.COCO17.index('rwri')
    assert ji.mirror_mapping[0] == 0


def test_config_structs_and_shipped_configs():
    from metrabs_amd.config import CONFIG_L_384, CONFIG_S_256, MetrabsConfig
    hp = CONFIG_S_256.head_params()
    assert (hp.proc_side, hp.stride_test, hp.centered_stride, hp.legacy_centered_stride_bug) == (256, 32, 0, 1)
    rp = CONFIG_L_384.recon_params()
    assert rp.proc_side == 384 and rp.mix_enabled == 1 and abs(rp.mix_3d_inside_fov - 0.5) < 1e-7
    assert abs(rp.l2_reg - 1e-2) < 1e-9 and abs(rp.weight_eps - 1e-4) < 1e-10
    assert MetrabsConfig().recon_params(mix_3d_inside_fov=None).mix_enabled == 0
    c = MetrabsConfig.from_any(cpu_ref.HeadConfig(proc_side=384, depth=4).as_dict())
    assert c.proc_side == 384 and c.depth == 4


def test_public_api_signature_matches_reference_defaults():
    """Argument names and defaults of the four public methods (multiperson_model.py:39-74,384-429)."""
    from metrabs_amd.multiperson.multiperson_model import Pose3dEstimator
    sig = inspect.signature(Pose3dEstimator.detect_poses)
    expected = dict(default_fov_degrees=55, internal_batch_size=64, antialias_factor=1, num_aug=5,
                    average_aug=True, skeleton='', detector_threshold=0.3,
                    detector_nms_iou_threshold=0.7, max_detections=-1, detector_flip_aug=False,
                    suppress_implausible_poses=True)
    for k, v in expected.items():
        assert sig.parameters[k].default == v, k
    names = list(inspect.signature(Pose3dEstimator.estimate_poses_batched).parameters)
    assert names == ['self', 'images', 'boxes', 'intrinsic_matrix', 'distortion_coeffs',
                     'extrinsic_matrix', 'world_up_vector', 'default_fov_degrees',
                     'internal_batch_size', 'antialias_factor', 'num_aug', 'average_aug', 'skeleton']
    from metrabs_amd.multiperson import multiperson_model as mm
    assert mm.UNKNOWN_INTRINSIC_MATRIX == ((-1, -1, -1),) * 3 and mm.DEFAULT_WORLD_UP == (0, -1, 0)


def test_warp_param_packing_and_host_distortion():
    from metrabs_amd.multiperson import warping
    from metrabs_amd.multiperson.multiperson_model import distort_points
    c = cases.warp_case('dist5')
    wp = warping.make_warp_params(c['K'], c['hinv'], c['dist'], c['crop_scales'], c['image_ids'])
    assert wp.shape == (6, 36)
    assert wp[:, 31].tolist() == cpu_ref.pyramid_level_index(c['crop_scales']).float().tolist()
    k1 = cpu_ref.corner_aligned_scale_mat(0.5) @ c['K'][2]
    assert torch.allclose(wp[2, 9:18].reshape(3, 3), k1)
    assert wp[:, 30].tolist() == [1.0] * 6 and wp[:, 33].tolist() == [1.0] * 6
    pts = torch.randn(6, 3, 17, 2, generator=cases.gen(3)) * 0.3
    d12 = warping.pad_axis_to_size(c['dist'], 12)
    assert torch.allclose(distort_points(pts, d12), cpu_ref.distort_points(pts, c['dist'][0]), atol=1e-7)
    assert torch.equal(distort_points(pts, torch.zeros(6, 12)), pts)


def test_cpu_tensors_are_rejected_without_fallback():
    from metrabs_amd import kernels
    from metrabs_amd.config import MetrabsConfig
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        kernels.softargmax_decode(torch.zeros(1, 9, 8, 8), 1, MetrabsConfig())
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        kernels.build_pyramid(torch.zeros(1, 3, 8, 8, dtype=torch.uint8))


def test_backbone_shapes():
    from metrabs_amd.backbones import build_backbone
    for name, res, c in [('resnet18', 128, 512), ('mobilenetv3', 128, 1280)]:
        net = build_backbone(name).eval()
        with torch.inference_mode():
            y = net(torch.rand(1, 3, res, res))
        assert y.shape == (1, c, res // 32, res // 32) and net.out_channels == c


def test_detector_geometry_matches_the_reference_arithmetic():
    """mtr_detector_geometry is host-only C (no GPU): person_detector.py:15-20,26-29 in float32 must
    give the oracle's numbers on every frame size, including the ones where float32 and float64
    would disagree about int(factor * w)."""
    import random
    from metrabs_amd import kernels
    from oracle import cpu_ref
    rnd = random.Random(0)
    sizes = [(1080, 1920), (1920, 1080), (270, 480), (416, 416), (100, 64), (37, 53), (512, 832),
             (2160, 3840), (1, 1), (415, 417)] + [(rnd.randint(8, 3000), rnd.randint(8, 4000))
                                                   for _ in range(2000)]
    import pytest
    for h, w in sizes:
        m = cpu_ref.detector_target_size(h, w)
        if min(m['target_h'], m['target_w']) <= 0:  # the reference's resize fails there too
            with pytest.raises(RuntimeError):
                kernels.detector_geometry(h, w)
            continue
        g = kernels.detector_geometry(h, w)
        assert (g.target_h, g.target_w, bool(g.antialias), g.pad_top, g.pad_left, g.out_h, g.out_w) == \
            (m['target_h'], m['target_w'], m['antialias'], m['pad_top'], m['pad_left'], m['out_h'],
             m['out_w']), (h, w)
        assert g.x_factor == m['x_factor'] and g.y_factor == m['y_factor'], (h, w)
        assert g.out_h % 32 == 0 and g.out_w % 32 == 0
    with pytest.raises(RuntimeError):
        kernels.detector_geometry(5000, 3)  # target width 0


def test_fold_batchnorm_is_the_same_function():
    """backbones.fold_batchnorm: every conv + BN pair of every backbone family becomes one conv with
    a bias; same features up to rounding; the original (checkpoint-compatible) network is untouched."""
    import torch
    from metrabs_amd import backbones
    for name, res in [('resnet18', 64), ('mobilenetv3', 64), ('effnetv2-s', 128)]:
        torch.manual_seed(0)
        # (enough samples per channel for sane running statistics: a random 40-layer network with
        #  degenerate statistics amplifies rounding differences chaotically)
        net = backbones.calibrate_batchnorm(backbones.build_backbone(name), res, 'cpu', batches=2, batch_size=4)
        keys = set(net.state_dict())
        folded = backbones.fold_batchnorm(net)
        assert set(net.state_dict()) == keys and net.out_channels == folded.out_channels
        assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in folded.modules())
        assert any(isinstance(m, backbones.DepthwiseConv2d) for m in folded.modules()) == (name != 'resnet18')
        x = torch.rand(2, 3, res, res, generator=torch.Generator().manual_seed(1))
        fused = backbones.fold_batchnorm(net, fused_epilogue=True)  # (CPU: its torch-op branch)
        assert any(isinstance(m, backbones.ConvBiasAct) for m in fused.modules())
        with torch.inference_mode():
            a, b, c = net(x), folded(x), fused(x)
        assert a.shape == b.shape == c.shape
        assert float((a - b).abs().max()) <= 2e-4 * float(a.abs().max()), name
        assert float((a - c).abs().max()) <= 2e-4 * float(a.abs().max()), name
    import pytest
    with pytest.raises(ValueError):
        backbones.fold_batchnorm(backbones.build_backbone('resnet18').train())


def test_bench_cli_contract_and_kernel_naming(monkeypatch):
    """bench.py: the driver's flags parse (--gpus N --steps K --warmup W, defaults N=1), and the
    roofline entry names the GEMM kernel mtr_head_fused dispatches to for the shape and precision."""
    import importlib
    import sys
    bench = importlib.import_module('bench')
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    a = bench.parse_args()
    assert (a.gpus, a.precision, a.no_fold_bn, a.no_fused_epilogue) == (1, 'f32', False, False)
    assert a.steps > 0 and a.warmup >= 0
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '7', '--warmup', '2'])
    a = bench.parse_args()
    assert (a.gpus, a.steps, a.warmup) == (8, 7, 2)
    # the roofline entry names the kernel the LIBRARY says it launches (mtr_head_plan, host-only)
    assert bench.head_kernel_name(64, 64, 17, 8) == 'head_rt_ld_kernel'              # config 2, f32: 3-tile blocks, loader wave
    assert bench.head_kernel_name(64, 1024, 17, 8) == 'head_rt_kernel'               # large launch: 5-tile blocks
    assert bench.head_kernel_name(64, 64, 17, 8, 'f32', 1283) == 'head_rt_kernel'    # C % 32 != 0: no loader kernel
    assert bench.head_kernel_name(64, 1024, 17, 72) == 'head_rt_kernel'              # D <= 80: 5-tile atoms
    assert bench.head_kernel_name(144, 32, 17, 8).startswith('head_rt_ld_kernel (+ head_rt_merge_kernel: 3 ')   # (round 6: 216 solo blocks of 4, 4, 2 tiles)
    assert bench.head_kernel_name(64, 64, 17, 8, 'f16', 1280) == 'head_fused16dma_kernel (early copies)'
    assert bench.head_kernel_name(36, 64, 17, 8, 'f16', 1280) == 'head_fused16_kernel'   # 6x6: registers
    assert bench.head_kernel_name(64, 64, 17, 8, 'f16', 1283).startswith('library')      # C % 8 != 0


def test_head_plan_follows_the_measured_model():
    """mtr_head_plan (host-only): the launch plan of the f32 head reproduces the measured best choice of
    the round-3 sweeps (profiles/r03*_head_sweep.jsonl, r03f_head_plan_check.jsonl) on their shapes."""
    from metrabs_amd import kernels
    p = kernels.head_plan(64, 1280, 8, 8, 17, 8)
    assert (p['kernel'], p['tiles_per_workgroup'], p['workgroups']) == ('head_rt_ld_kernel', 3, 256)
    p = kernels.head_plan(8, 1280, 8, 8, 17, 8)
    assert (p['kernel'], p['tiles_per_workgroup']) == ('head_rt_ld_kernel', 1)
    # 12x12 maps = 2 blocks of 64 positions + 16: the last blocks of 4 consecutive crops share a workgroup
    # (32 crops x 5 blocks of 2 tiles x 2.25 column blocks = 360 workgroups instead of 480)
    p = kernels.head_plan(32, 1280, 12, 12, 17, 8)
    assert (p['kernel'], p['tiles_per_workgroup'], p['split_column_blocks'], p['workgroups']) == \
        ('head_rt_ld_kernel', 4, 3, 216)   # (round 6: 2-tile blocks paired in a single round are 2.2 x, not 1.8 x: measured)
    p = kernels.head_plan(32, 1280, 12, 12, 122, 8)   # configs[4]'s head in f32: 69 row tiles
    assert (p['kernel'], p['tiles_per_workgroup'], p['split_column_blocks'], p['workgroups']) == \
        ('head_rt_kernel', 5, 3, 32 * 14 * 2 + 8 * 14)
    assert kernels.head_plan(32, 1280, 12, 12, 17, 8, have_workspace=False)['split_column_blocks'] == 0
    p = kernels.head_plan(64, 1280, 16, 16, 17, 8)
    assert (p['kernel'], p['tiles_per_workgroup'], p['split_column_blocks']) == ('head_rt_kernel', 5, 4)
    p = kernels.head_plan(1024, 1280, 8, 8, 17, 8)
    assert (p['kernel'], p['tiles_per_workgroup'], p['split_column_blocks']) == ('head_rt_kernel', 5, 0)
    # the launch is simulated (blocks of r, r, ..., rest tiles over the CUs' slots): 320 crops are 960 blocks
    # of 4, 4, 2 tiles rather than 640 of 5 (1.25 rounds); a 24x24 map's 9 column blocks go to blocks of 2
    p = kernels.head_plan(320, 1280, 8, 8, 17, 8)
    assert (p['kernel'], p['tiles_per_workgroup'], p['workgroups']) == ('head_rt_kernel', 4, 960)
    assert 80.0 < p['model_us'] < 120.0   # (measured: 91 - 95 us)
    p = kernels.head_plan(16, 1280, 24, 24, 17, 8)
    assert (p['kernel'], p['tiles_per_workgroup'], p['split_column_blocks']) == ('head_rt_kernel', 2, 9)
    for b, tiles in ((96, 2), (128, 5), (160, 4), (192, 2), (256, 5), (512, 5), (4096, 5), (40000, 5)):
        assert kernels.head_plan(b, 1280, 8, 8, 17, 8)['tiles_per_workgroup'] == tiles, b
    assert kernels.head_plan(64, 1280, 8, 8, 17, 72)['tiles_per_workgroup'] == 5     # a 72-bin joint = one atom
    assert kernels.head_plan(64, 1280, 8, 8, 17, 8, rt_k_groups=2, rt_loader=1)['kernel'] == 'head_rt_ks_kernel'
    assert kernels.head_plan(64, 1280, 7, 7, 17, 8) is None                          # H*W % 4 != 0: library path


def test_head_options_struct_is_versioned_by_its_size():
    """mtr_head_options.struct_size (include/metrabs_hip.h): the library reads the fields inside the caller's
    size and takes its own choice for those behind it; a size of 0 (a zeroed struct, or a round-3 caller whose
    struct began with rt_tiles_per_workgroup) is MTR_E_PARAM, never a misread.  Checked on the host-only
    mtr_head_plan."""
    import ctypes
    from metrabs_amd import _lib
    lib = _lib.load()

    def plan(opts):
        info = _lib.HeadPlanInfo()
        rc = lib.mtr_head_plan(_lib.MTR_F32, _lib.MTR_NCHW, 64, 1280, 8, 8, 17, 8, ctypes.byref(opts), 1,
                               ctypes.byref(info))
        return rc, info.kernel, info.tiles_per_workgroup

    full = _lib.head_options(rt_tiles=2, rt_loader=1)
    assert full.struct_size == ctypes.sizeof(_lib.HeadOptions) == 32
    rc, kernel, tiles = plan(full)
    assert rc == 0 and tiles == 2 and _lib.HEAD_KERNEL_NAMES[kernel] == 'head_rt_kernel'
    short = _lib.head_options(rt_tiles=2, rt_loader=1)
    short.struct_size = 8          # an older caller that only knows rt_tiles_per_workgroup
    rc, kernel, tiles = plan(short)
    assert rc == 0 and tiles == 2  # rt_loader (behind its size) is the library's choice, not the 1 in memory
    assert (kernel, tiles) == plan(_lib.head_options(rt_tiles=2))[1:]
    for bad in (0, 3, 6, 260):
        zero = _lib.head_options()
        zero.struct_size = bad
        assert plan(zero)[0] != 0
    longer = (ctypes.c_uint32 * 16)(64, 2)   # a FUTURE caller with a longer struct: the known prefix is read
    info = _lib.HeadPlanInfo()
    rc = lib.mtr_head_plan(_lib.MTR_F32, _lib.MTR_NCHW, 64, 1280, 8, 8, 17, 8,
                           ctypes.cast(longer, ctypes.POINTER(_lib.HeadOptions)), 1, ctypes.byref(info))
    assert rc == 0 and info.tiles_per_workgroup == 2


def test_head_plan_reads_the_batch_size_in_eights_and_the_auto_rule_not_at_all():
    import torch
    from metrabs_amd import kernels
    plans = [kernels.head_plan(B, 1280, 8, 8, 17, 8) for B in range(57, 65)]
    assert all(p == plans[0] for p in plans)
    assert kernels.head_plan(65, 1280, 8, 8, 17, 8)['workgroups'] != plans[0]['workgroups']   # (72 crops: another plan)
    # MetrabsHeads(fused='auto'): a static rule without the batch size
    import inspect
    assert 'B' not in inspect.signature(kernels.head_auto_choice).parameters
    assert kernels.head_auto_choice(1280, 17, 8, 8, 8) and kernels.head_auto_choice(1280, 17, 16, 8, 8)
    # f32 beyond 16 depth bins (the metric string's 72 included): the library pair is faster, 'auto' yields
    assert not kernels.head_auto_choice(1280, 17, 72, 8, 8) and not kernels.head_auto_choice(1280, 17, 24, 8, 8)
    assert kernels.head_auto_choice(1280, 17, 72, 8, 8, dtype=torch.float16)
    assert kernels.head_auto_choice(1280, 122, 8, 12, 12, dtype=torch.float16)
    assert not kernels.head_auto_choice(1280, 17, 8, 24, 24)
    # (round 6) the layout is an input: f32 channels_last maps of more than 64 positions take the library GEMM (F.linear)
    assert kernels.head_auto_choice(1280, 17, 8, 8, 8, True) and kernels.head_auto_choice(1280, 17, 8, 12, 12)
    assert not kernels.head_auto_choice(1280, 17, 8, 12, 12, True) and not kernels.head_auto_choice(1280, 17, 72, 8, 8, True)
    assert kernels.head_auto_choice(1280, 122, 8, 12, 12, True, torch.float16)
    assert not kernels.head_auto_choice(1280, 17, 8, 20, 20, dtype=torch.bfloat16)
    assert not kernels.head_auto_choice(1280, 17, 81, 8, 8)   # no fused kernel beyond 80 depth bins


def test_metrabs_affine_latent_modes_follow_the_reference_constructor(tmp_path):
    """Row a11, host side (models/metrabs.py:23-44): raw point counts per mode, the affine-weights file /
    name / dict forms, 'affine weights not used', and which rows of conv_final a latent prefix selects."""
    import os
    from metrabs_amd.config import MetrabsConfig
    from metrabs_amd.joint_info import JointInfo
    from metrabs_amd.models.metrabs import Metrabs, MetrabsHeads, load_affine_weights
    J, n_lat = 17, 12
    ji = JointInfo([f'j{i}' for i in range(J)], [[0, 1]])
    w1 = np.random.RandomState(0).rand(J, n_lat).astype(np.float32)
    w2 = np.random.RandomState(1).rand(n_lat, J).astype(np.float32)
    path = str(tmp_path / 'aff.npz')
    np.savez(path, w1=w1, w2=w2)
    os.makedirs(tmp_path / 'skeleton_conversion')
    np.savez(str(tmp_path / 'skeleton_conversion' / 'named.npz'), w1=w1, w2=w2)
    for mode, n_raw, out_latent, prefix in (('transform_coords', n_lat, True, None),
                                            ('predict_all_and_latents', n_lat + J, True, n_lat),
                                            ('regularize_to_manifold', J, False, None)):
        m = Metrabs(torch.nn.Identity(), ji, MetrabsConfig(affine_weights=path, **{mode: True}), in_channels=8)
        assert (m.n_raw_points, m.latent_output, m.latent_prefix, m.n_latents) == (n_raw, out_latent, prefix, n_lat)
        assert m.heatmap_heads.conv_final.out_channels == n_raw * 9
        assert torch.equal(m.reconstruction_weights, torch.from_numpy(w1) @ torch.from_numpy(w2))
        assert 'recombination_weights' not in m.state_dict()   # plain attributes in the reference
    os.environ['DATA_ROOT'], old = str(tmp_path), os.environ.get('DATA_ROOT')
    try:
        a, b = load_affine_weights('named')   # $DATA_ROOT/skeleton_conversion/<name>.npz (:24-25)
    finally:
        os.environ.pop('DATA_ROOT') if old is None else os.environ.__setitem__('DATA_ROOT', old)
    assert a.shape == (J, n_lat) and b.shape == (n_lat, J)
    with pytest.raises(ValueError, match='affine weights not used'):
        Metrabs(torch.nn.Identity(), ji, MetrabsConfig(), in_channels=8, affine_weights=dict(w1=w1, w2=w2))
    with pytest.raises(ValueError):
        Metrabs(torch.nn.Identity(), ji, MetrabsConfig(transform_coords=True), in_channels=8)
    with pytest.raises(ValueError):   # w2 maps to another joint count
        Metrabs(torch.nn.Identity(), JointInfo(['a', 'b'], [[0, 1]]), MetrabsConfig(transform_coords=True),
                in_channels=8, affine_weights=dict(w1=w1, w2=w2))
    plain = Metrabs(torch.nn.Identity(), ji, MetrabsConfig(), in_channels=8)
    assert plain.n_raw_points == J and not plain.latent_output and plain.n_latents is None
    # the rows of a latent prefix: j < k of the 2D block and of every depth slice (channel J + d*J + j, :79)
    heads = MetrabsHeads(5, MetrabsConfig(depth=2), in_channels=4)
    assert heads._point_rows(2, 'cpu').tolist() == [0, 1, 5, 6, 10, 11]
    wsel, bsel = heads._weights(2)
    assert torch.equal(wsel, heads.conv_final.weight.detach()[[0, 1, 5, 6, 10, 11], :, 0, 0])
    assert heads._weights(2)[0] is wsel                      # cached per weight version ...
    with torch.no_grad():
        heads.conv_final.bias.add_(1.0)
    assert heads._weights(2)[0] is not wsel                  # ... and rebuilt after an in-place edit
    snap = heads.packed_snapshot()
    assert heads.snapshot_is_current(snap)
    with torch.no_grad():
        heads.conv_final.weight.mul_(2.0)
    assert not heads.snapshot_is_current(snap)
    gen0 = plain.storage_generation
    plain.double()
    assert plain.storage_generation > gen0


def test_head_plan_takes_the_weights_in_registers_kernel_on_large_wide_launches():
    """Round 5 (csrc/head_areg.hip): the library's own choice for >= 512 crops of >= 8 joint groups on 5 column
    tiles (J = 122 on 12x12, 16-bit) -- 7 % ahead of the early-copies kernel at 1024 crops, behind it at 256
    (profiles/r05m_areg_frag.jsonl); dma_staging 4 forces it wherever it exists, with 2 - 4 waves per workgroup."""
    from metrabs_amd import kernels
    name = 'head_fused16areg_kernel (weights in registers)'
    big = kernels.head_plan(1024, 1280, 12, 12, 122, 8, torch.float16)
    assert big['kernel'] == name and big['tiles_per_workgroup'] == 4 and big['workgroups'] == 1024 * 5
    assert kernels.head_plan(256, 1280, 12, 12, 122, 8, torch.float16)['kernel'] != name     # configs[4]'s sizes:
    assert kernels.head_plan(32, 1280, 12, 12, 122, 8, torch.float16)['kernel'] != name      # the early-copies kernel
    assert kernels.head_plan(1024, 1280, 12, 12, 17, 8, torch.float16)['kernel'] != name     # 3 joint groups
    assert kernels.head_plan(1024, 1280, 8, 8, 122, 8, torch.float16)['kernel'] != name      # 2 column tiles
    forced = kernels.head_plan(32, 1280, 12, 12, 122, 8, torch.bfloat16, dma_staging=4, groups_per_workgroup=2)
    assert forced['kernel'] == name and forced['tiles_per_workgroup'] == 2 and forced['workgroups'] == 32 * 9
    assert kernels.head_plan(32, 1280, 8, 8, 17, 8, torch.float16, dma_staging=4)['kernel'] != name   # no such tile: as -1
    # the blob carries the fragment-major copy of the joint-group weights wherever that kernel exists
    lib = kernels._lib.load()
    assert lib.mtr_head_packed_bytes(1280, 122, 8, 1) - lib.mtr_head_packed_bytes(1288, 122, 8, 1) > 18 * 20 * 8192


def test_the_determinism_pin_is_held_while_any_thread_is_inside_a_backbone_call():
    """ torch.backends.cudnn.deterministic is process-global: the pin is a lock-protected depth counter,
    so one thread leaving its backbone call cannot switch the flag off under another thread's running call, and
    the caller's own setting is what comes back at the end."""
    import threading
    from metrabs_amd.models.metrabs import _deterministic_convolutions as pin
    before = torch.backends.cudnn.deterministic
    try:
        for found in (False, True):
            torch.backends.cudnn.deterministic = found
            inside_a, release_a, seen = threading.Event(), threading.Event(), {}

            def thread_a():
                with pin():
                    inside_a.set()
                    release_a.wait(10)
                    seen['a_at_exit'] = torch.backends.cudnn.deterministic

            t = threading.Thread(target=thread_a)
            t.start()
            assert inside_a.wait(10)
            with pin():   # thread B enters and leaves while A is still inside
                assert torch.backends.cudnn.deterministic
            assert torch.backends.cudnn.deterministic, 'B leaving must not unpin A'
            release_a.set()
            t.join(10)
            assert seen['a_at_exit'] is True
            assert torch.backends.cudnn.deterministic is found   # the caller's own setting is back
    finally:
        torch.backends.cudnn.deterministic = before


def test_predict_multi_is_strict_about_the_tf_signature():
    """metrabs_tf/models/metrabs.py:71-73: float16 [N, H, W, 3] crops and float32 [N, 3, 3] intrinsics; anything
    else is a TypeError (tf.function's input_signature), CPU tensors are rejected like everywhere (no fallback)."""
    from metrabs_amd.joint_info import JointInfo
    from metrabs_amd.models.metrabs import Metrabs
    m = Metrabs(torch.nn.Identity(), JointInfo(cases.COCO17, cases.COCO17_EDGES), in_channels=8)
    K = torch.eye(3)[None].repeat(2, 1, 1)
    with pytest.raises(TypeError):
        m.predict_multi(torch.zeros(2, 64, 64, 3), K)                       # f32 crops
    with pytest.raises(TypeError):
        m.predict_multi(torch.zeros(2, 3, 64, 64, dtype=torch.float16), K)  # NCHW crops
    with pytest.raises(TypeError):
        m.predict_multi(torch.zeros(2, 64, 64, 3, dtype=torch.float16), K.double())
    with pytest.raises(TypeError):
        m.predict_multi(torch.zeros(2, 64, 64, 3, dtype=torch.float16), K[:1])
    with pytest.raises((RuntimeError, ValueError, TypeError)):              # CPU tensors: no CPU path
        m.predict_multi(torch.zeros(2, 64, 64, 3, dtype=torch.float16), K)


def test_head_plan_takes_one_group_per_workgroup_on_a_tight_stage_where_the_round_model_says_so():
    """Round 6: the 16-bit early-copies kernel with one joint group per workgroup on a feature stage of exactly H*W
    positions (52 KiB of LDS at 12x12: three workgroups per CU) -- the library's choice where its model of the launch's
    resident rounds promises >= 7 % over two groups per workgroup (profiles/r06l_head16_tight_crossover.jsonl: 32 and 64
    crops of J = 122 yes, 48 / 96 / 160 / 256 no), never for other map sizes, few joint groups or forced options."""
    from metrabs_amd import _lib, kernels
    tight, early = _lib.HEAD_KERNEL_NAMES[18], _lib.HEAD_KERNEL_NAMES[14]
    plan = lambda B, J=122, side=12, **kw: kernels.head_plan(B, 1280, side, side, J, 8, torch.float16, **kw)
    for layout in (False, True):
        assert plan(32, channels_last=layout)['kernel'] == tight
        assert plan(32, channels_last=layout)['tiles_per_workgroup'] == 1 and plan(32, channels_last=layout)['workgroups'] == 32 * 18
        assert plan(64, channels_last=layout)['kernel'] == tight
        for B in (48, 96, 160, 256):
            assert plan(B, channels_last=layout)['kernel'] == early, B
        assert plan(8, channels_last=layout)['kernel'] == tight            # (one group either way: one more slot per CU)
    assert plan(128, J=60)['kernel'] == tight and plan(256, J=60)['kernel'] == tight    # 9 joint groups
    assert plan(32, J=17)['kernel'] != tight                                              # 3 joint groups: the old rule
    assert plan(32, side=8)['kernel'] != tight and plan(32, side=16)['kernel'] != tight   # other map sizes
    assert plan(32, dma_staging=3)['kernel'] == early and plan(32, groups_per_workgroup=2)['kernel'] == early
    forced = plan(256, dma_staging=7, groups_per_workgroup=2)
    assert forced['kernel'] == tight and forced['tiles_per_workgroup'] == 2
    assert plan(512)['kernel'] == _lib.HEAD_KERNEL_NAMES[15]                              # large launches: weights in registers
