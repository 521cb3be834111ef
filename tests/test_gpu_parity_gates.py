"""The north-star parity definition -- poses3d from IDENTICAL features within 1e-3 mm of the
metrabs_pytorch CPU path -- as HARD gates at every BASELINE.json config shape, in three regimes.

* 'consistent_low' / 'consistent_peaked': features + default-initialised conv_final whose logits
  describe a plausible pose (cases.consistent_head_case: Gaussian bumps of height 4, |logit| <= 5,
  or height 25 around each joint; the person fills the crop 2.5 - 4.5 m from the camera).
  Gate: MPJPE(ours, fp64) <= 5e-4 mm in both; MPJPE(ours, oracle) <= 1e-3 mm in the low regime and
  <= 1.5e-3 mm in the peaked one (where the oracle itself sits 7e-4 .. 1e-3 mm from fp64) -- fixed
  numbers, with the documented exceptions of CONSISTENT_BOUND where the oracle is farther still.
* 'random_head': N(0,1) features x default-initialised conv_final x 8 (logits +-25), the kind of
  input bench.py's random network produces.  Its heatmaps are nearly uniform, all joints decode to
  the crop centre and the reference-point depth -- the ratio of two vanishing spreads -- is
  ill-conditioned (median depth ~0 mm): the reference's own fp32 result is 1e-3 ... 4e-3 mm from an
  fp64 evaluation of the same formulas, so no implementation can sit within 1e-3 mm of it.  Gates
  there: FIXED numbers per case (~2x the values measured when they were set,
  profiles/r02j_parity_report.jsonl) on ours-vs-fp64 and on ours-vs-oracle, so that a head that
  gets noisier fails whatever the oracle's own floor does.

f16 features (configs[4]): the oracle evaluates the f32 conv on the same rounded features and
weights (products of two f16 values are exact in f32), so the same gates apply.
Every case appends its numbers to gpurun_out/parity_report.jsonl (kept under profiles/ per round).
"""
import json
import os

import pytest
import torch

from conftest import ROOT
from oracle import cases, cpu_ref

pytestmark = pytest.mark.gpu

# name: (B, C, J, map side, proc_side, depth bins, feature dtype)
SHAPES = {
    'configs[0] ResNet-18 256 B=1': (1, 512, 17, 8, 256, 8, torch.float32),
    'configs[1] EffNetV2-S 256 B=64': (64, 1280, 17, 8, 256, 8, torch.float32),
    'configs[2] EffNetV2-L 384 B=32/GPU': (32, 1280, 17, 12, 384, 8, torch.float32),
    'configs[2] EffNetV2-L 384 B=256 on one GPU': (256, 1280, 17, 12, 384, 8, torch.float32),
    'configs[3] MobileNetV3 256, 8 boxes x 5 aug': (40, 1280, 17, 8, 256, 8, torch.float32),
    'configs[4] EffNetV2-L 384 f16 J=122 B=32/GPU': (32, 1280, 122, 12, 384, 8, torch.float16),
    'metric string: 72 depth bins, 256 px, B=64': (64, 1280, 17, 8, 256, 72, torch.float32),
}

# mm: (MPJPE ours-vs-fp64, MPJPE ours-vs-oracle, max-abs ours-vs-oracle); measured r02b (re-measured r02j) (profiles/
# r02j_parity_report.jsonl): ours-vs-fp64 2.7e-4 .. 6.6e-4, ours-vs-oracle 7.3e-4 .. 1.5e-3 (= the
# oracle's own 6.7e-4 .. 1.3e-3 from fp64), max 2.9e-3 .. 5.9e-3
RANDOM_HEAD_BOUNDS = {
    'configs[0] ResNet-18 256 B=1': (6e-4, 3e-3, 6e-3),
    'configs[1] EffNetV2-S 256 B=64': (7e-4, 2e-3, 1e-2),
    'configs[2] EffNetV2-L 384 B=32/GPU': (7e-4, 1.5e-3, 6e-3),
    'configs[2] EffNetV2-L 384 B=256 on one GPU': (7e-4, 1.5e-3, 7e-3),
    'configs[3] MobileNetV3 256, 8 boxes x 5 aug': (7e-4, 2e-3, 8e-3),
    'configs[4] EffNetV2-L 384 f16 J=122 B=32/GPU': (1.3e-3, 2e-3, 1.2e-2),
    'metric string: 72 depth bins, 256 px, B=64': (7e-4, 2.2e-3, 1.2e-2),
}
# consistent heads, MPJPE ours-vs-oracle: 1e-3 mm everywhere (measured 4.4e-4 .. 9.0e-4) except the
# two cases where the ORACLE's own fp32 result is farther than that from fp64:
#  * configs[0] is ONE crop: MPJPE over its 17 joints is the error of ONE reference point, and the
#    oracle's fp32 LAPACK lstsq lands 6e-4 .. 2.1e-3 mm from fp64 on this crop depending on the host
#    CPU of the box (three runs of the round: low regime 1.9e-3 / 5.9e-4 / 5.9e-4, peaked 9.7e-4 /
#    9.7e-4 / 2.1e-3) while ours stays at 1.1e-4 / 1.5e-4 from fp64 in every run;
#  * 72 depth bins, peaked: 1,241 output rows on K = 1280 make the minimum-norm features large
#    (std 40): the oracle's conv is 1.5e-3 mm from fp64 (ours: 5.3e-4).
CONSISTENT_BOUND = {('configs[0] ResNet-18 256 B=1', 'consistent_low'): 3e-3,
                    ('configs[0] ResNet-18 256 B=1', 'consistent_peaked'): 3e-3,
                    ('metric string: 72 depth bins, 256 px, B=64', 'consistent_peaked'): 2.5e-3}
# ... and MPJPE ours-vs-fp64: 5e-4 mm everywhere (measured 1.2e-4 .. 3.3e-4), 1e-3 for the second one
CONSISTENT_FP64_BOUND = {('metric string: 72 depth bins, 256 px, B=64', 'consistent_peaked'): 1e-3}


def make_inputs(name, regime):
    B, C, J, hw, P, D, dtype = SHAPES[name]
    seed = 9100 + sum(ord(c) for c in name)
    if regime == 'random_head':
        g = cases.gen(seed)
        feat = torch.randn(B, C, hw, hw, generator=g)
        w, b = cases.default_conv_init(J * (1 + D), C, g)
        w, b = w * 8.0, b * 8.0
        f = (450 + 100 * torch.rand(B, generator=g)) * P / 256
        K = torch.zeros(B, 3, 3)
        K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = f, f, P / 2, P / 2, 1
    else:
        amp = 4.0 if regime == 'consistent_low' else 25.0
        feat, w, b, K = cases.consistent_head_case(B, C, J, hw, P, D, amp, seed)
    return feat.to(dtype), w, b, K


def run_case(name, regime):
    """-> dict of distances (mm) between ours (HIP fused head + reconstruct through the C-ABI), the
    oracle (fp32 CPU restatement of the reference) and an fp64 evaluation, on identical features."""
    from metrabs_amd import kernels
    from metrabs_amd.config import MetrabsConfig
    B, C, J, hw, P, D, dtype = SHAPES[name]
    feat, w, b, K = make_inputs(name, regime)
    ocfg = cpu_ref.HeadConfig(proc_side=P, depth=D)
    cfg = MetrabsConfig(proc_side=P, depth=D)
    with torch.inference_mode():
        wk = cases.head_weights_as_consumed(w, dtype)
        ref = cpu_ref.crop_model_from_features(feat.float(), wk, b, K, J, ocfg)
        truth = cpu_ref.crop_model_from_features_fp64(feat.float(), wk, b, K, J, ocfg)
        logits_absmax = float(torch.nn.functional.conv2d(feat.float(), wk[:, :, None, None], b).abs().max())
    assert kernels.head_fused_supported(C, J, D, hw, hw, dtype=dtype)
    packed = kernels.head_pack_weights(w.cuda(), b.cuda(), J, D, dtype)
    c2d, c3d = kernels.head_fused(feat.cuda(), packed, C, J, cfg)
    ours = kernels.reconstruct_absolute(c2d, c3d, K.cuda(), cfg).cpu()
    assert torch.isfinite(ours).all()
    r = dict(case=name, regime=regime, logits_absmax=round(logits_absmax, 2),
             median_depth_mm=round(float(truth[..., 2].median()), 1),
             mpjpe_ours_vs_ref=cpu_ref.mpjpe(ours, ref), max_ours_vs_ref=float((ours - ref).abs().max()),
             mpjpe_ours_vs_fp64=cpu_ref.mpjpe(ours, truth),
             max_ours_vs_fp64=float((ours.double() - truth).abs().max()),
             mpjpe_ref_vs_fp64=cpu_ref.mpjpe(ref, truth),
             max_ref_vs_fp64=float((ref.double() - truth).abs().max()))
    out_dir = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, 'parity_report.jsonl'), 'a') as fh:
            fh.write(json.dumps(r) + '\n')
    print('[parity]', json.dumps(r))
    return r


@pytest.mark.parametrize('regime', ['consistent_low', 'consistent_peaked'])
@pytest.mark.parametrize('name', list(SHAPES))
def test_plausible_poses_are_within_1e3_mm_of_the_reference(name, regime, hip_lib):
    r = run_case(name, regime)
    # low regime: the north-star 1e-3 mm (measured 4.4e-4 .. 7.2e-4).  Peaked regime: measured
    # 6.9e-4 .. 9.1e-4 = the oracle's OWN distance to fp64 (6.7e-4 .. 9.7e-4, host-CPU dependent):
    # fixed 1.5e-3 so that another box's oneDNN / LAPACK code path cannot fail it, with the hard
    # ours-vs-fp64 bound below doing the real work
    bound = CONSISTENT_BOUND.get((name, regime), 1e-3 if regime == 'consistent_low' else 1.5e-3)
    assert r['median_depth_mm'] > 1500  # a person in front of the camera, not a degenerate solve
    assert (r['logits_absmax'] <= 5.5) if regime == 'consistent_low' else (r['logits_absmax'] >= 20)
    assert r['mpjpe_ours_vs_ref'] <= bound, r
    assert r['mpjpe_ours_vs_fp64'] <= CONSISTENT_FP64_BOUND.get((name, regime), 5e-4), r


@pytest.mark.parametrize('name', list(SHAPES))
def test_random_head_fixed_bounds(name, hip_lib):
    r = run_case(name, 'random_head')
    b64, bref, bmax = RANDOM_HEAD_BOUNDS[name]
    assert r['logits_absmax'] >= 15.0
    assert r['mpjpe_ours_vs_fp64'] <= b64, r
    assert r['mpjpe_ours_vs_ref'] <= bref, r
    assert r['max_ours_vs_ref'] <= bmax, r
