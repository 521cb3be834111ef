"""GPU: Metrabs.predict_multi(image f16 [N,res,res,3], intrinsic_matrix f32 [N,3,3]) -> poses3d f32 [N,J,3] -- the
bare-bones crop-model entry of the TF twin (metrabs_tf/models/metrabs.py:71-78, docs/INFERENCE.md:112-132; SURVEY
section 3.2) -- against the STORED output of the reference on the same rounded features (the configs[4] parity-gate
goldens: the reference's MetrabsHeads.forward + reconstruct_absolute on f16 features), <= 1e-3 mm MPJPE.

The backbone is a stand-in that hands the golden's features back in the TF twin's layout (NHWC memory, f16), so
what is gated is predict_multi's own path: the interleaved crops reach the backbone as an NCHW view of the caller's
memory (no layout copy), 16-bit autocast, the NHWC 16-bit fused head consuming the features in place,
reconstruction."""
import pytest
import torch

from conftest import load_golden
from oracle import cases, cpu_ref

pytestmark = pytest.mark.gpu


class _Injected(torch.nn.Module):
    def __init__(self, features_nhwc):
        super().__init__()
        self.features = features_nhwc                      # [B, C, h, w] view over NHWC memory
        self.out_channels = features_nhwc.shape[1]
        self.seen = None

    def forward(self, image):
        self.seen = dict(shape=tuple(image.shape), dtype=image.dtype, ptr=image.data_ptr(),
                         channels_last=image.is_contiguous(memory_format=torch.channels_last),
                         autocast=torch.is_autocast_enabled(), autocast_dtype=torch.get_autocast_gpu_dtype())
        return self.features


@pytest.mark.parametrize('regime', ['consistent_low', 'consistent_peaked'])
def test_predict_multi_on_injected_f16_nhwc_features_within_1e3_mm(regime, hip_lib):
    from metrabs_amd.config import MetrabsConfig
    from metrabs_amd.joint_info import JointInfo
    from metrabs_amd.models.metrabs import Metrabs
    name = 'configs[4] EffNetV2-L 384 f16 J=122 B=32/GPU'
    B, C, J, hw, P, D, dtype = cases.PARITY_GATE_SHAPES[name]
    feat, w, b, K = cases.parity_gate_inputs(name, regime)
    g = load_golden(cases.parity_gate_slug(name, regime))
    ref, truth = torch.from_numpy(g['poses3d']), torch.from_numpy(g['poses3d_fp64'])
    feats = feat.cuda().contiguous(memory_format=torch.channels_last)
    backbone = _Injected(feats)
    names = [f'j{i}' for i in range(J)]
    model = Metrabs(backbone, JointInfo(names, []), MetrabsConfig(proc_side=P, depth=D), in_channels=C)
    with torch.no_grad():
        model.heatmap_heads.conv_final.weight.copy_(w[:, :, None, None])
        model.heatmap_heads.conv_final.bias.copy_(b)
    model = model.cuda().eval()
    image = torch.rand(B, P, P, 3, generator=cases.gen(5)).half().cuda()
    with torch.inference_mode():
        poses = model.predict_multi(image, K.cuda())
    assert poses.shape == (B, J, 3) and poses.dtype == torch.float32
    s = backbone.seen
    assert s['shape'] == (B, 3, P, P) and s['dtype'] == torch.float16
    assert s['ptr'] == image.data_ptr() and s['channels_last'], 'the crops reach the backbone in place, interleaved'
    assert s['autocast'] and s['autocast_dtype'] == torch.float16
    assert model.heatmap_heads.last_path == 'fused'
    ours = poses.cpu()
    err, err64 = cpu_ref.mpjpe(ours, ref), cpu_ref.mpjpe(ours, truth)
    print(f'[parity] predict_multi {regime}: MPJPE vs reference {err:.2e} mm, vs fp64 {err64:.2e} mm')
    assert err <= 1e-3 and err64 <= 5e-4
    # the same bits as the NCHW entry on the same features and intrinsics (forward under the same autocast)
    with torch.inference_mode():
        again = model.forward((image.permute(0, 3, 1, 2), K.cuda()), autocast_dtype=torch.float16)
    assert torch.equal(again, poses)


def test_predict_multi_through_a_real_convolutional_backbone(hip_lib):
    """End to end with convolutions in the loop (the e2e tiny backbone under f16 autocast): interleaved f16 crops
    in, finite f32 poses out, the head on 16-bit features.  (No numeric gate here: the tiny random head's
    reference depth is ill-conditioned, f16 against f32 arithmetic moves it by metres; the gate is the test above.)"""
    from test_gpu_e2e import build_estimator
    case = cases.e2e_case('aug5')
    est = build_estimator(case, 'auto')
    model = est.crop_model
    res = case['res']
    crops = torch.rand(6, res, res, 3, generator=cases.gen(11)).half().cuda()
    K = cases.intrinsics_for(res, res, 40.0)[None].repeat(6, 1, 1).cuda()
    seen = {}
    hook = model.heatmap_heads.register_forward_pre_hook(lambda m, args: seen.update(dtype=args[0].dtype))
    with torch.inference_mode():
        p16 = model.predict_multi(crops, K)
    hook.remove()
    assert p16.shape == (6, 17, 3) and p16.dtype == torch.float32 and torch.isfinite(p16).all()
    assert seen['dtype'] == torch.float16
