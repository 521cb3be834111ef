"""TEST INFRASTRUCTURE: deterministic synthetic inputs for the oracle / golden vectors / parity tests.

Inputs follow SURVEY.md section 8(d): seeded torch.Generator (CPU RNG -- deterministic for a fixed
torch build; every golden file also stores a sha256 of its regenerated inputs so that an RNG change
is detected instead of silently comparing different data).
"""
import hashlib

import numpy as np
import torch
import torch.nn.functional as F

from oracle.cpu_ref import HeadConfig

COCO17 = ['nose', 'leye', 'reye', 'lear', 'rear', 'lsho', 'rsho', 'lelb', 'relb', 'lwri', 'rwri',
          'lhip', 'rhip', 'lkne', 'rkne', 'lank', 'rank']
COCO17_EDGES = [(0, 1), (0, 2), (1, 3), (2, 4), (5, 6), (5, 7), (7, 9), (6, 8), (8, 10), (5, 11),
                (6, 12), (11, 12), (11, 13), (13, 15), (12, 14), (14, 16)]


def mirror_mapping(names):
    """Left/right swap by leading 'l'/'r' (posepile JointInfo convention; see
    oracle/ref_harness.py:_JointInfoStub)."""
    index = {n: i for i, n in enumerate(names)}
    out = []
    for n in names:
        if n.startswith('l') and ('r' + n[1:]) in index:
            out.append(index['r' + n[1:]])
        elif n.startswith('r') and ('l' + n[1:]) in index:
            out.append(index['l' + n[1:]])
        else:
            out.append(index[n])
    return np.array(out, dtype=np.int64)


def sha256_of(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        a = t.detach().cpu().contiguous().numpy() if isinstance(t, torch.Tensor) else np.ascontiguousarray(t)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def gen(seed):
    return torch.Generator().manual_seed(int(seed))


# ------------------------------------------------------------------------------------ heads

HEAD_CASES = {
    # name: dict(B, J, D, H, W, sigma, dtype, cfg overrides)
    's256': dict(B=4, J=17, D=8, H=8, W=8, sigma=1.0, seed=101, cfg={}),
    's256_legacy': dict(B=4, J=17, D=8, H=8, W=8, sigma=3.0, seed=102,
                        cfg=dict(centered_stride=False, legacy_centered_stride_bug=True)),
    's256_peaked': dict(B=4, J=17, D=8, H=8, W=8, sigma=10.0, seed=103, cfg={}),
    'l384': dict(B=3, J=17, D=8, H=12, W=12, sigma=1.0, seed=104, cfg=dict(proc_side=384)),
    'l384_j122': dict(B=1, J=122, D=8, H=12, W=12, sigma=2.0, seed=105, cfg=dict(proc_side=384)),
    's256_d72': dict(B=2, J=17, D=72, H=8, W=8, sigma=1.0, seed=106, cfg=dict(depth=72)),
    's256_stride16': dict(B=2, J=17, D=8, H=16, W=16, sigma=2.0, seed=107,
                          cfg=dict(stride_test=16)),
    'odd160': dict(B=3, J=5, D=8, H=5, W=5, sigma=2.0, seed=108, cfg=dict(proc_side=160)),
    's256_fp16': dict(B=4, J=17, D=8, H=8, W=8, sigma=3.0, seed=109, dtype='float16', cfg={}),
    's256_j1': dict(B=2, J=1, D=8, H=8, W=8, sigma=1.0, seed=110, cfg={}),
    's256_spike': dict(B=2, J=17, D=8, H=8, W=8, sigma=1.0, seed=111, spike=60.0, cfg={}),
}


def head_case(name):
    c = dict(HEAD_CASES[name])
    cfg = HeadConfig(**c['cfg'])
    g = gen(c['seed'])
    n_out = c['J'] * (1 + c['D'])
    logits = torch.randn(c['B'], n_out, c['H'], c['W'], generator=g) * c['sigma']
    if c.get('spike'):
        # one dominant voxel per joint: exercises the exp range of the softmax
        for b in range(c['B']):
            for j in range(c['J']):
                d = int(torch.randint(0, c['D'], (1,), generator=g))
                h = int(torch.randint(0, c['H'], (1,), generator=g))
                w = int(torch.randint(0, c['W'], (1,), generator=g))
                logits[b, c['J'] + d * c['J'] + j, h, w] += c['spike']
                logits[b, j, h, w] += c['spike']
    if c.get('dtype') == 'float16':
        logits = logits.half()
    return logits, c['J'], cfg


HEADCONV_CASES = {
    's256_c64': dict(B=4, C=64, J=17, D=8, H=8, W=8, gain=1.0, seed=201, cfg={}),
    's256_c1280': dict(B=2, C=1280, J=17, D=8, H=8, W=8, gain=1.0, seed=202, cfg={}),
    's256_c1280_peaked': dict(B=2, C=1280, J=17, D=8, H=8, W=8, gain=20.0, seed=203, cfg={}),
    'l384_c1280': dict(B=2, C=1280, J=17, D=8, H=12, W=12, gain=4.0, seed=204,
                       cfg=dict(proc_side=384)),
    'r18_c512': dict(B=1, C=512, J=17, D=8, H=8, W=8, gain=4.0, seed=205, cfg={}),
    'l384_j122_c96': dict(B=2, C=96, J=122, D=8, H=12, W=12, gain=4.0, seed=206,
                          cfg=dict(proc_side=384)),
}


def default_conv_init(n_out, c_in, g):
    """torch.nn.Conv2d default init (kaiming_uniform a=sqrt(5) -> U(-1/sqrt(fan_in), ..))."""
    bound = 1.0 / np.sqrt(c_in)
    w = (torch.rand(n_out, c_in, generator=g) * 2 - 1) * bound
    b = (torch.rand(n_out, generator=g) * 2 - 1) * bound
    return w, b


def headconv_case(name):
    c = dict(HEADCONV_CASES[name])
    cfg = HeadConfig(**c['cfg'])
    g = gen(c['seed'])
    feat = torch.randn(c['B'], c['C'], c['H'], c['W'], generator=g)
    w, b = default_conv_init(c['J'] * (1 + c['D']), c['C'], g)
    return feat, w * c['gain'], b * c['gain'], c['J'], cfg


def consistent_head_case(B, C, J, hw, proc_side, D, amp, seed, spread=0.18):
    """Features + head parameters whose logits describe a PLAUSIBLE pose (a person filling most of
    the crop, 2.5 - 4.5 m from the camera), for the features -> poses3d parity gates.

    Why not random features x random weights: their heatmaps are nearly uniform, every joint decodes
    to the crop centre, the 2D and 3D spreads both vanish and the reference-point depth (their
    ratio, ptu3d.py:56-105) is ill-conditioned -- the median depth of such a batch is ~0 mm and even
    exactly rounded logits land 2e-3 mm from an fp64 evaluation.  No detector crop looks like that.

    Construction: conv_final keeps torch's default initialisation (W [N, C], bias); the target
    logits L* are Gaussian bumps of height `amp` (4: |logit| <= 5, 'low'; 25: 'peaked') around each
    joint's 2D / 3D heatmap position; the features of every position are the minimum-norm solution
    of W f = L* - bias plus a random N(0,1) vector projected onto W's null space, computed in
    float64 and rounded to float32 once.  So the features look like noise of unit scale (the
    K = C accumulation sees realistic magnitudes), the logits equal L* to rounding, and the poses
    are well conditioned.  -> (features [B,C,hw,hw] f32, weight [N,C], bias [N], K [B,3,3])"""
    g = gen(seed)
    w, b = default_conv_init(J * (1 + D), C, g)
    w, b = w.double(), b.double()
    f = (450 + 100 * torch.rand(B, generator=g)) * proc_side / 256
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = f, f, proc_side / 2, proc_side / 2, 1
    rel = (torch.randn(B, J, 3, generator=g, dtype=torch.float64) *
           torch.tensor([0.13, 0.17, 0.12], dtype=torch.float64)).clamp(-0.3, 0.3)
    u3 = 0.5 + rel                                   # (x, y, z) in heatmap units
    u2 = 0.5 + rel[..., :2] / (1.0 + 0.5 * rel[..., 2:])  # nearer joints project farther out
    gx = torch.linspace(0, 1, hw, dtype=torch.float64)
    gz = torch.linspace(0, 1, D, dtype=torch.float64)
    d2 = ((gx[None, None, None, :] - u2[..., 0, None, None]) ** 2 +
          (gx[None, None, :, None] - u2[..., 1, None, None]) ** 2)
    l2 = amp * torch.exp(-d2 / (2 * spread * spread))                                 # [B,J,h,w]
    d3 = ((gx[None, None, None, None, :] - u3[..., 0, None, None, None]) ** 2 +
          (gx[None, None, None, :, None] - u3[..., 1, None, None, None]) ** 2 +
          (gz[None, None, :, None, None] - u3[..., 2, None, None, None]) ** 2)
    l3 = amp * torch.exp(-d3 / (2 * spread * spread))                                 # [B,J,D,h,w]
    # conv_final's channel order: J 2D rows, then slice d of joint j at J + d*J + j
    target = torch.cat([l2, l3.permute(0, 2, 1, 3, 4).reshape(B, D * J, hw, hw)], dim=1)
    target = (target - b[None, :, None, None]).reshape(B, J * (1 + D), hw * hw)
    w_pinv = w.T @ torch.linalg.inv(w @ w.T)         # [C, N]
    noise = torch.randn(B, C, hw * hw, generator=g, dtype=torch.float64)
    feat = w_pinv @ target + noise - w_pinv @ (w @ noise)
    return feat.reshape(B, C, hw, hw).float(), w.float(), b.float(), K


def consistent_head_for_features(features, J, D, proc_side, amp, seed, spread=0.18, ridge=1e-6):
    """The converse of consistent_head_case for GIVEN features (a real backbone's output, [B,C,h,w] with
    B*h*w <= C): conv_final parameters under which THESE features describe a plausible pose -- the same
    Gaussian-bump target logits, W = T F^T (F F^T + ridge * mean diag)^-1 in float64 (a ridge keeps the
    weights small when the backbone's features are nearly dependent; the logits then equal the targets
    only approximately, which is all that is asked: peaked heatmaps, a person 2.5 - 4.5 m away), bias 0.
    -> (weight [N, C] f32, bias [N] f32, K [B,3,3])"""
    B, C, hw, hw2 = features.shape
    assert hw == hw2 and B * hw * hw <= C
    g = gen(seed)
    f = (450 + 100 * torch.rand(B, generator=g)) * proc_side / 256
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = f, f, proc_side / 2, proc_side / 2, 1
    rel = (torch.randn(B, J, 3, generator=g, dtype=torch.float64) *
           torch.tensor([0.13, 0.17, 0.12], dtype=torch.float64)).clamp(-0.3, 0.3)
    u3 = 0.5 + rel
    u2 = 0.5 + rel[..., :2] / (1.0 + 0.5 * rel[..., 2:])
    gx = torch.linspace(0, 1, hw, dtype=torch.float64)
    gz = torch.linspace(0, 1, D, dtype=torch.float64)
    d2 = ((gx[None, None, None, :] - u2[..., 0, None, None]) ** 2 +
          (gx[None, None, :, None] - u2[..., 1, None, None]) ** 2)
    l2 = amp * torch.exp(-d2 / (2 * spread * spread))
    d3 = ((gx[None, None, None, None, :] - u3[..., 0, None, None, None]) ** 2 +
          (gx[None, None, None, :, None] - u3[..., 1, None, None, None]) ** 2 +
          (gz[None, None, :, None, None] - u3[..., 2, None, None, None]) ** 2)
    l3 = amp * torch.exp(-d3 / (2 * spread * spread))
    target = torch.cat([l2, l3.permute(0, 2, 1, 3, 4).reshape(B, D * J, hw, hw)], dim=1)   # [B, N, h, w]
    T = target.permute(1, 0, 2, 3).reshape(J * (1 + D), B * hw * hw)                        # [N, B*HW]
    F = features.double().permute(1, 0, 2, 3).reshape(C, B * hw * hw)                        # [C, B*HW]
    gram = F.T @ F                                                                           # [BHW, BHW]
    gram = gram + ridge * gram.diagonal().mean() * torch.eye(gram.shape[0], dtype=torch.float64)
    w = torch.linalg.solve(gram, T.T).T @ F.T                                                # [N, C]
    return w.float(), torch.zeros(J * (1 + D)), K


# ---- the features -> poses3d parity gates (tests/test_gpu_parity_gates.py, bench.py's parity probe):
# every BASELINE.json config shape and the metric string's 72 depth bins.
# name: (B, C, J, map side, proc_side, depth bins, feature dtype)
PARITY_GATE_SHAPES = {
    'configs[0] ResNet-18 256 B=1': (1, 512, 17, 8, 256, 8, torch.float32),
    'configs[1] EffNetV2-S 256 B=64': (64, 1280, 17, 8, 256, 8, torch.float32),
    'configs[2] EffNetV2-L 384 B=32/GPU': (32, 1280, 17, 12, 384, 8, torch.float32),
    'configs[2] EffNetV2-L 384 B=256 on one GPU': (256, 1280, 17, 12, 384, 8, torch.float32),
    'configs[3] MobileNetV3 256, 8 boxes x 5 aug': (40, 1280, 17, 8, 256, 8, torch.float32),
    'configs[4] EffNetV2-L 384 f16 J=122 B=32/GPU': (32, 1280, 122, 12, 384, 8, torch.float16),
    'metric string: 72 depth bins, 256 px, B=64': (64, 1280, 17, 8, 256, 72, torch.float32),
}
PARITY_GATE_REGIMES = ('consistent_low', 'consistent_peaked', 'random_head')


def parity_gate_slug(name, regime):
    key = {'configs[0] ResNet-18 256 B=1': 'cfg0_r18_b1', 'configs[1] EffNetV2-S 256 B=64': 'cfg1_s_b64',
           'configs[2] EffNetV2-L 384 B=32/GPU': 'cfg2_l_b32', 'configs[2] EffNetV2-L 384 B=256 on one GPU': 'cfg2_l_b256',
           'configs[3] MobileNetV3 256, 8 boxes x 5 aug': 'cfg3_mnv3_b40',
           'configs[4] EffNetV2-L 384 f16 J=122 B=32/GPU': 'cfg4_l_f16_j122_b32',
           'metric string: 72 depth bins, 256 px, B=64': 'metric_d72_b64'}[name]
    return f'parity_{key}_{regime}'


def parity_gate_inputs(name, regime):
    """-> (features in the shape's dtype, weight [N,C] f32, bias, K [B,3,3]); seeded, CPU.
    consistent_low / consistent_peaked: consistent_head_case with bumps of 4 / 25; random_head: N(0,1)
    features x default-initialised conv_final x 8 (logits +-25) -- what a random-weight network emits."""
    B, C, J, hw, P, D, dtype = PARITY_GATE_SHAPES[name]
    seed = 9100 + sum(ord(c) for c in name)
    if regime == 'random_head':
        g = gen(seed)
        feat = torch.randn(B, C, hw, hw, generator=g)
        w, b = default_conv_init(J * (1 + D), C, g)
        w, b = w * 8.0, b * 8.0
        f = (450 + 100 * torch.rand(B, generator=g)) * P / 256
        K = torch.zeros(B, 3, 3)
        K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = f, f, P / 2, P / 2, 1
    else:
        amp = 4.0 if regime == 'consistent_low' else 25.0
        feat, w, b, K = consistent_head_case(B, C, J, hw, P, D, amp, seed)
    return feat.to(dtype), w, b, K


# ---- row a11: the affine-latent options of Metrabs (models/metrabs.py:23-44,52-62)
# name: (mode, B, C, n_joints, n_latents, map side, proc_side, depth bins, bump height, seed)
LATENT_CASES = {
    'transform_b8': ('transform_coords', 8, 256, 17, 12, 8, 256, 8, 4.0, 9501),
    'all_and_latents_b8': ('predict_all_and_latents', 8, 512, 17, 12, 8, 256, 8, 25.0, 9502),
    'all_and_latents_j60_384': ('predict_all_and_latents', 4, 1280, 60, 32, 12, 384, 8, 4.0, 9503),
    'regularize_b4': ('regularize_to_manifold', 4, 256, 17, 12, 8, 256, 8, 4.0, 9504),
}


def affine_weights_case(n_joints, n_latents, seed):
    """A synthetic affine-weights file: w1 [J, n_latents] (every latent point an affine combination of
    the joints: its column sums to 1) and w2 [n_latents, J] (every joint an affine combination of the
    latent points, negative weights included) -- what the reference's skeleton_conversion/*.npz hold."""
    g = gen(seed)
    w1 = torch.softmax(2.0 * torch.randn(n_joints, n_latents, generator=g), dim=0)
    r = 0.3 * torch.randn(n_latents, n_joints, generator=g)
    w2 = r - r.mean(dim=0, keepdim=True) + 1.0 / n_latents
    return w1.float().contiguous(), w2.float().contiguous()


def latent_case(name):
    """-> dict(cfg, features [B,C,h,w], weight [N,C], bias [N], K [B,3,3], w1, w2, n_joints, n_latents,
    n_raw): a plausible-pose head (consistent_head_case) over the model's RAW points."""
    mode, B, C, J, n_lat, hw, P, D, amp, seed = LATENT_CASES[name]
    cfg = HeadConfig(proc_side=P, depth=D, **{mode: True})
    n_raw = {'transform_coords': n_lat, 'predict_all_and_latents': n_lat + J,
             'regularize_to_manifold': J}[mode]
    feat, w, b, K = consistent_head_case(B, C, n_raw, hw, P, D, amp, seed)
    w1, w2 = affine_weights_case(J, n_lat, seed + 50)
    return dict(cfg=cfg, features=feat, weight=w, bias=b, K=K, w1=w1, w2=w2, n_joints=J, n_latents=n_lat,
                n_raw=n_raw)


# ------------------------------------------------------------------------------------ reconstruct

RECON_CASES = {
    'b64_j17': dict(B=64, J=17, seed=301, cfg={}),
    'b5_j122_384': dict(B=5, J=122, seed=302, cfg=dict(proc_side=384)),
    'b1_j17': dict(B=1, J=17, seed=303, cfg={}),
    'b8_legacy': dict(B=8, J=17, seed=304,
                      cfg=dict(centered_stride=False, legacy_centered_stride_bug=True)),
    'b8_weak': dict(B=8, J=17, seed=305, cfg=dict(weak_perspective=True)),
    'b6_nomix': dict(B=6, J=17, seed=306, cfg=dict(mix_3d_inside_fov=None)),
    'b4_outfov': dict(B=4, J=17, seed=307, cfg={}, all_out_of_fov=True),
}


def recon_case(name):
    """Synthetic *consistent* pose (SURVEY.md 8c KAT 5): pick ref point, K and rel pose, project,
    add prediction noise; ~10 % joints pushed out of the FOV."""
    c = dict(RECON_CASES[name])
    cfg_kw = dict(c['cfg'])
    mix = cfg_kw.pop('mix_3d_inside_fov', 0.5) if 'mix_3d_inside_fov' in cfg_kw else 0.5
    cfg = HeadConfig(**cfg_kw)
    cfg.mix_3d_inside_fov = mix
    g = gen(c['seed'])
    B, J, P = c['B'], c['J'], cfg.proc_side
    f = (450 + 100 * torch.rand(B, generator=g)) * (P / 256)
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0] = f
    K[:, 1, 1] = f * (0.98 + 0.04 * torch.rand(B, generator=g))
    K[:, 0, 2] = P / 2
    K[:, 1, 2] = P / 2
    K[:, 2, 2] = 1
    ref = torch.stack([
        200 * torch.randn(B, generator=g), 200 * torch.randn(B, generator=g),
        2000 + 3000 * torch.rand(B, generator=g)], dim=1)
    rel = torch.randn(B, J, 3, generator=g) * torch.tensor([350.0, 450.0, 250.0])
    abs3d = rel + ref[:, None]
    proj = abs3d[..., :2] / abs3d[..., 2:]
    coords2d = proj * torch.stack([K[:, 0, 0], K[:, 1, 1]], dim=1)[:, None] + K[:, None, :2, 2]
    coords2d = coords2d + 1.5 * torch.randn(B, J, 2, generator=g)
    rel = rel + 15 * torch.randn(B, J, 3, generator=g)
    push = torch.rand(B, J, generator=g) < 0.1
    coords2d = torch.where(push[..., None], coords2d * 0.02 - 3.0, coords2d)
    if c.get('all_out_of_fov'):
        coords2d[0] = -50.0 + torch.rand(J, 2, generator=g)
    return coords2d.contiguous(), rel.contiguous(), K, cfg


# ------------------------------------------------------------------------------------ images / warp

def weak_perspective_kat():
    """Hand-derived known answer for reconstruct_absolute(weak_perspective=True) (ptu3d.py:9-49,
    ptu.py:4-34) at the default config (FOV bounds [24, 232] px, mix 0.5), K = [[500,0,128],
    [0,500,128],[0,0,1]].

    Crop 0: four in-FOV joints on a rectangle + one joint outside the FOV (masked out of every
    mean).  Normalised 2D: x in {-0.04, 0.08}, y in {-0.12, 0.04} -> mean (0.02, -0.04), deviations
    (+-0.06, +-0.08): stdev2d = sqrt(4 (0.0036 + 0.0064) / 4) = 0.1.  Relative 3D: x in {-120, 180},
    y in {-220, 180} -> mean (30, -20), deviations (+-150, +-200): stdev3d = sqrt(4 (22500 + 40000) / 4)
    = 250; z mean 50.  Reference depth = 250 / 0.1 = 2500, ref = (0.02, -0.04, 1) 2500 - (30, -20, 50)
    = (20, -80, 2450).  In-FOV joints: 0.5 (rel + ref) + 0.5 (x_n, y_n, 1) (rel_z + 2450); the masked
    joint: rel + ref.
    Crop 1: no joint in the FOV -> masked means are nan_to_num(0 / 0) = 0, both stdevs
    max(sqrt(0 + 1e-10), 1e-5) -> depth 1, ref = (0, 0, 1); every joint: rel + (0, 0, 1).
    -> (coords2d [2,5,2], coords3d_rel [2,5,3], K [2,3,3], expected poses3d [2,5,3])"""
    K = torch.tensor([[500.0, 0, 128], [0, 500.0, 128], [0, 0, 1]]).repeat(2, 1, 1)
    c2d = torch.tensor([[[108.0, 68.0], [168.0, 68.0], [108.0, 148.0], [168.0, 148.0], [10.0, 128.0]],
                        [[5.0, 5.0], [250.0, 5.0], [5.0, 250.0], [250.0, 250.0], [128.0, 240.0]]])
    rel = torch.tensor([[[-120.0, -220.0, -100.0], [180.0, -220.0, 50.0], [-120.0, 180.0, 150.0],
                         [180.0, 180.0, 100.0], [999.0, -999.0, 999.0]],
                        [[10.0, 20.0, 30.0], [-40.0, 50.0, -60.0], [70.0, -80.0, 90.0],
                         [0.0, 0.0, 0.0], [-1.0, 2.0, -3.0]]])
    want = torch.tensor([[[-97.0, -291.0, 2350.0], [200.0, -300.0, 2500.0], [-102.0, 102.0, 2600.0],
                          [202.0, 101.0, 2550.0], [1019.0, -1079.0, 3449.0]],
                         [[10.0, 20.0, 31.0], [-40.0, 50.0, -59.0], [70.0, -80.0, 91.0],
                          [0.0, 0.0, 1.0], [-1.0, 2.0, -2.0]]])
    return c2d, rel, K, want


def box_consistency_kat():
    """Hand-derived known answers for is_pose_consistent_with_box (TF plausibility_check.py:66-84;
    PyTorch port :86-107): the pose's 2D bounding box is (10,10)-(50,90).
    -> (pose2d [6,3,2], boxes [6,5], expected bool [6])"""
    pose = torch.tensor([[10.0, 90.0], [50.0, 10.0], [30.0, 40.0]])
    rows = [
        ([0.0, 0.0, 60.0, 100.0, 1.0], True),     # intersection 40 x 80 = 3200 > 3000
        ([0.0, 0.0, 100.0, 100.0, 1.0], False),   # 3200 < 5000
        ([200.0, 0.0, 50.0, 50.0, 1.0], False),   # disjoint: relu -> 0
        ([10.0, 10.0, 80.0, 80.0, 0.3], False),   # 40 x 80 = 3200 > 3200 is false (strict >)
        ([30.0, 50.0, 40.0, 80.0, 0.9], False),   # x 30..50, y 50..90: 800 < 1600
        ([20.0, 20.0, 20.0, 60.0, 0.9], True),    # box inside the pose box: 1200 > 600
    ]
    boxes = torch.tensor([r[0] for r in rows])
    want = torch.tensor([r[1] for r in rows])
    return pose[None].repeat(len(rows), 1, 1), boxes, want


def synth_images(n, h, w, seed):
    """uint8 [n,3,h,w]: even images are uniform noise (worst case for bilinear rounding), odd
    images are smooth gradients + mild noise."""
    g = gen(seed)
    imgs = torch.randint(0, 256, (n, 3, h, w), dtype=torch.uint8, generator=g)
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    for i in range(1, n, 2):
        base = torch.stack([
            127 + 100 * torch.sin(xx / 17.0 + i), 127 + 100 * torch.cos(yy / 23.0 - i),
            (xx + yy).float() % 256], dim=0)
        noise = torch.randint(-6, 7, (3, h, w), generator=g)
        imgs[i] = torch.clip(base + noise, 0, 255).to(torch.uint8)
    return imgs


def synth_boxes(n_images, h, w, max_boxes, seed, min_boxes=0):
    """list of [n_i,5] float32 boxes (x, y, w, h, conf=1); may contain empty entries."""
    g = gen(seed)
    out = []
    for _ in range(n_images):
        n = int(torch.randint(min_boxes, max_boxes + 1, (1,), generator=g))
        bw = (0.15 + 0.35 * torch.rand(n, generator=g)) * w
        bh = (0.3 + 0.6 * torch.rand(n, generator=g)) * h
        bx = torch.rand(n, generator=g) * (w - bw * 0.7) - 0.15 * bw
        by = torch.rand(n, generator=g) * (h - bh * 0.7) - 0.15 * bh
        out.append(torch.stack([bx, by, bw, bh, torch.ones(n)], dim=1).float())
    return out


DISTORTION_5 = (-0.1, 0.01, 1e-3, 1e-3, 0.0)
DISTORTION_12 = (-0.08, 0.012, 8e-4, -6e-4, 1e-3, 0.01, -2e-3, 1e-4, 3e-4, -2e-4, 1e-4, 2e-4)


def intrinsics_for(h, w, fov_degrees=55.0, jitter_seed=None):
    f = max(h, w) / (np.tan(np.deg2rad(fov_degrees) / 2) * 2)
    K = torch.tensor([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], dtype=torch.float32)
    if jitter_seed is not None:
        g = gen(jitter_seed)
        K[0, 0] *= float(0.95 + 0.1 * torch.rand(1, generator=g))
        K[1, 1] *= float(0.95 + 0.1 * torch.rand(1, generator=g))
        K[0, 1] = float(0.5 * torch.randn(1, generator=g))  # a little skew
        K[0, 2] += float(3 * torch.randn(1, generator=g))
        K[1, 2] += float(3 * torch.randn(1, generator=g))
    return K


# ------------------------------------------------------------------------------------ e2e tiny model

class TinyBackbone(torch.nn.Module):
    """Deterministic stand-in for the CNN backbone (out of scope): /32
    average pool, 1x1 conv 3->C, tanh.  Cheap, and nearly order-independent in fp32 so that CPU
    and GPU features agree to ~1e-6."""

    def __init__(self, c_out, stride, seed):
        super().__init__()
        g = gen(seed)
        self.stride = stride
        self.proj = torch.nn.Conv2d(3, c_out, 1)
        with torch.no_grad():
            self.proj.weight.copy_(torch.randn(c_out, 3, 1, 1, generator=g) * 2.0)
            self.proj.bias.copy_(torch.randn(c_out, generator=g) * 0.5)

    def forward(self, image):
        x = F.avg_pool2d(image.float(), self.stride)
        return torch.tanh(self.proj(x * 2 - 1)) * 2.0


def tiny_head_weights(c_in, n_points, depth, seed, gain=6.0):
    g = gen(seed)
    w, b = default_conv_init(n_points * (1 + depth), c_in, g)
    return w * gain, b * gain


class PlausiblePoseBackbone(torch.nn.Module):
    """Stand-in backbone whose features, under the head of `plausible_pose_model`, describe a person
    filling most of the crop 1.5 - 4 m away -- for estimator tests that compare two EXECUTION ORDERS
    of the same crops (ranks, slices, graphs) and must not sit on the x / z singularity a random head
    produces (tests/test_gpu_sharded_estimator.py).

    The pose is a function of the CROP alone: twelve colour statistics of the crop (quadrant means per
    channel, float64, quantised to 2^-12 so that a reduction-order difference cannot move them) perturb
    a fixed base pose; the target logits are consistent_head_case's Gaussian bumps; the head is an
    matrix Q [N, C] of orthonormal rows (C = 160 >= N channels), so features = Q^T (target - bias),
    evaluated in float64 and rounded to float32 once: a crop's features do not depend on the batch it
    travels in."""

    def __init__(self, n_joints, depth, hw, seed, channels=160, amp=6.0, spread=0.18, zoom=1.5):
        super().__init__()
        g = gen(seed)
        n = n_joints * (1 + depth)
        assert channels >= n
        q, _ = torch.linalg.qr(torch.randn(channels, n, generator=g, dtype=torch.float64))
        self.n_joints, self.depth, self.hw, self.amp, self.spread, self.zoom = n_joints, depth, hw, amp, spread, zoom
        self.out_channels = channels
        self.register_buffer('q', q.T.contiguous())                                       # head weight [N, C]
        self.register_buffer('bias', (torch.rand(n, generator=g, dtype=torch.float64) * 2 - 1) * 0.05)
        base = (torch.randn(n_joints, 3, generator=g, dtype=torch.float64) *
                torch.tensor([0.13, 0.17, 0.12], dtype=torch.float64)).clamp(-0.25, 0.25)
        self.register_buffer('base', base)
        self.register_buffer('mix', torch.randn(12, n_joints * 3, generator=g, dtype=torch.float64))

    def head_parameters(self):
        return self.q.float(), self.bias.float()

    def forward(self, image):
        B, J, D, hw = image.shape[0], self.n_joints, self.depth, self.hw
        x = image.double()
        h2, w2 = x.shape[2] // 2, x.shape[3] // 2
        stats = torch.stack([x[:, :, :h2, :w2].mean((2, 3)), x[:, :, :h2, w2:].mean((2, 3)),
                             x[:, :, h2:, :w2].mean((2, 3)), x[:, :, h2:, w2:].mean((2, 3))], dim=2).reshape(B, 12)
        stats = torch.round(stats * 4096) / 4096
        rel = (self.base[None] + 0.06 * torch.tanh(3 * (stats - 0.3) @ self.mix).reshape(B, J, 3)).clamp(-0.3, 0.3)
        u3 = 0.5 + rel
        u2 = 0.5 + self.zoom * rel[..., :2] / (1.0 + 0.5 * rel[..., 2:])
        gx = torch.linspace(0, 1, hw, dtype=torch.float64, device=x.device)
        gz = torch.linspace(0, 1, D, dtype=torch.float64, device=x.device)
        s2 = 2 * self.spread * self.spread
        d2 = ((gx[None, None, None, :] - u2[..., 0, None, None]) ** 2 +
              (gx[None, None, :, None] - u2[..., 1, None, None]) ** 2)
        l2 = self.amp * torch.exp(-d2 / s2)
        d3 = ((gx[None, None, None, None, :] - u3[..., 0, None, None, None]) ** 2 +
              (gx[None, None, None, :, None] - u3[..., 1, None, None, None]) ** 2 +
              (gz[None, None, :, None, None] - u3[..., 2, None, None, None]) ** 2)
        l3 = self.amp * torch.exp(-d3 / s2)
        target = torch.cat([l2, l3.permute(0, 2, 1, 3, 4).reshape(B, D * J, hw, hw)], dim=1)
        target = target - self.bias[None, :, None, None]
        feat = torch.einsum('nc,bnp->bcp', self.q, target.reshape(B, -1, hw * hw))
        return feat.reshape(B, -1, hw, hw).float()


def with_plausible_pose_model(case, seed=977):
    """An e2e_case whose crop model is PlausiblePoseBackbone + its orthonormal-row head (C = 160 channels)."""
    cfg = case['cfg']
    bb = PlausiblePoseBackbone(17, cfg.depth, cfg.proc_side // cfg.stride_test, seed)
    w, b = bb.head_parameters()
    return dict(case, backbone=bb, head_w=w, head_b=b, C=bb.out_channels)


E2E_CASES = {
    # res 64 keeps the golden crops small; geometry/TTA/post-processing code paths are identical.
    'aug1': dict(seed=401, n_images=2, imh=120, imw=160, res=64, num_aug=1, aa=1, dist=None,
                 ibs=64, average_aug=True, extr=False, skeleton=False, jtm=False, known_k=False),
    'aug5': dict(seed=402, n_images=3, imh=120, imw=160, res=64, num_aug=5, aa=1, dist=None,
                 ibs=10, average_aug=True, extr=False, skeleton=False, jtm=False, known_k=True),
    'aug5_dist_aa2': dict(seed=403, n_images=2, imh=150, imw=200, res=64, num_aug=5, aa=2,
                          dist=DISTORTION_5, ibs=64, average_aug=False, extr=True, skeleton=True,
                          jtm=True, known_k=True),
    'aug4_dist12': dict(seed=404, n_images=2, imh=120, imw=160, res=64, num_aug=4, aa=1,
                        dist=DISTORTION_12, ibs=3, average_aug=True, extr=True, skeleton=False,
                        jtm=False, known_k=True),
    'aug2_aa8': dict(seed=409, n_images=3, imh=160, imw=200, res=32, num_aug=2, aa=8, dist=DISTORTION_5,
                     ibs=64, average_aug=True, extr=False, skeleton=False, jtm=False, known_k=True),
    'aug2_aa4_bigbox': dict(seed=405, n_images=1, imh=400, imw=600, res=32, num_aug=2, aa=4,
                            dist=None, ibs=64, average_aug=True, extr=False, skeleton=False,
                            jtm=False, known_k=False),
}

E2E_C = 24  # tiny backbone channels
# cases whose backbone outputs are stored too (golden e2efeat_*): the estimator's glue without the sampler
E2E_FEATURE_CASES = ('aug5', 'aug4_dist12', 'aug1', 'aug5_dist_aa2')
E2E_SKELETON = [0, 5, 6, 11, 12, 15, 16]


def e2e_case(name):
    c = dict(E2E_CASES[name])
    images = synth_images(c['n_images'], c['imh'], c['imw'], c['seed'])
    boxes = synth_boxes(c['n_images'], c['imh'], c['imw'], 3, c['seed'] + 1, min_boxes=1)
    if name == 'aug5':
        boxes[1] = boxes[1][:0]  # an image without detections (ragged edge case)
    if c['known_k']:
        K = torch.stack([intrinsics_for(c['imh'], c['imw'], 60.0, c['seed'] + 10 + i)
                         for i in range(c['n_images'])])
    else:
        K = torch.tensor([[[-1.0] * 3] * 3])
    dist = torch.tensor([list(c['dist'])], dtype=torch.float32) if c['dist'] else torch.zeros(1, 5)
    if c['extr']:
        g = gen(c['seed'] + 2)
        ang = 0.3 * torch.randn(3, generator=g)
        cx, sx, cy, sy = torch.cos(ang[0]), torch.sin(ang[0]), torch.cos(ang[1]), torch.sin(ang[1])
        rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        E = torch.eye(4)
        E[:3, :3] = rx @ ry
        E[:3, 3] = torch.tensor([100.0, -50.0, 800.0])
        extr = E[None]
        world_up = torch.tensor([0.0, 0.0, 1.0])
    else:
        extr = torch.eye(4)[None]
        world_up = torch.tensor([0.0, -1.0, 0.0])
    # stride res/8 -> an 8x8 heatmap, the EffNetV2-S/256 shape, at a tiny crop resolution
    cfg = HeadConfig(proc_side=c['res'], stride_train=c['res'] // 8, stride_test=c['res'] // 8)
    backbone = TinyBackbone(E2E_C, cfg.stride_test, c['seed'] + 3)
    w, b = tiny_head_weights(E2E_C, 17, cfg.depth, c['seed'] + 4)
    jtm = None
    if c['jtm']:
        g = gen(c['seed'] + 5)
        jtm = torch.eye(17) * 0.7 + 0.3 * torch.softmax(torch.randn(17, 17, generator=g), dim=0)
    skeleton = E2E_SKELETON if c['skeleton'] else list(range(17))
    return dict(images=images, boxes=boxes, K=K, dist=dist, extr=extr, world_up=world_up,
                cfg=cfg, backbone=backbone, head_w=w, head_b=b, jtm=jtm, skeleton=skeleton,
                num_aug=c['num_aug'], aa=c['aa'], ibs=c['ibs'], average_aug=c['average_aug'],
                res=c['res'])


WARP_CASES = {
    'nodist': dict(seed=501, dist=None, res=48),
    'dist5': dict(seed=502, dist=DISTORTION_5, res=48),
    'dist12': dict(seed=503, dist=DISTORTION_12, res=40),
}


def warp_case(name):
    """Direct inputs of warping.warp_images_with_pyramid: 2 images, 6 crops spanning all three
    pyramid levels and partially out-of-frame homographies."""
    c = dict(WARP_CASES[name])
    g = gen(c['seed'])
    imh, imw, res = 120, 160, c['res']
    images_u8 = synth_images(2, imh, imw, c['seed'])
    images = (images_u8.float() / 255) ** 2.2
    n = 6
    K = torch.stack([intrinsics_for(imh, imw, 55.0, c['seed'] + i) for i in range(n)])
    crop_scales = torch.tensor([1.6, 0.9, 0.45, 0.3, 0.2, 0.1])
    image_ids = torch.tensor([0, 1, 0, 1, 1, 0])
    hinvs = []
    for i in range(n):
        s = float(crop_scales[i])
        newK = torch.tensor([[K[i, 0, 0] * s, K[i, 0, 1] * s, res / 2],
                             [0, K[i, 1, 1] * s, res / 2], [0, 0, 1]])
        ang = 0.4 * torch.randn(3, generator=g)
        ca, sa = torch.cos(ang[2]), torch.sin(ang[2])
        rz = torch.tensor([[ca, -sa, 0], [sa, ca, 0], [0, 0, 1]])
        cb, sb = torch.cos(ang[0] * 0.3), torch.sin(ang[0] * 0.3)
        ry = torch.tensor([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
        hinvs.append(torch.linalg.inv(newK @ (rz @ ry)))
    hinv = torch.stack(hinvs)
    dist = (torch.tensor([list(c['dist'])] * n, dtype=torch.float32) if c['dist']
            else torch.zeros(n, 5))
    return dict(images_u8=images_u8, images=images, K=K, hinv=hinv, dist=dist,
                crop_scales=crop_scales, image_ids=image_ids, res=res)


# ------------------------------------------------------------------------- detector pre-processing

# name -> (n_images, h, w, seed): shrinking (antialiased) landscape / portrait / odd sizes, exact fit,
# enlarging (plain bilinear), and a size whose target is already a multiple of 32 (no padding)
DETPRE_CASES = {
    'qhd_270x480': (2, 270, 480, 21),
    'portrait_320x180': (1, 320, 180, 22),
    'odd_211x307': (2, 211, 307, 23),
    'exact_416x416': (1, 416, 416, 24),
    'small_100x64': (2, 100, 64, 25),
    'tiny_37x53': (1, 37, 53, 26),
    'nopad_512x832': (1, 512, 832, 27),
}


def detpre_case(name):
    n, h, w, seed = DETPRE_CASES[name]
    images = synth_images(n, h, w, seed)
    g = gen(seed + 1000)
    # detector answers in the padded network frame (x1, y1, x2, y2, conf); image 0 may be empty
    boxes = []
    for i in range(n):
        k = int(torch.randint(0 if i else 1, 5, (1,), generator=g))
        x1 = torch.rand(k, generator=g) * 300
        y1 = torch.rand(k, generator=g) * 300
        boxes.append(torch.stack([x1, y1, x1 + 20 + 100 * torch.rand(k, generator=g),
                                  y1 + 30 + 100 * torch.rand(k, generator=g),
                                  torch.rand(k, generator=g)], dim=1).float())
    return dict(images=images, net_boxes=boxes)


# ------------------------------------------------------------------ plausibility filter + pose NMS

# a standing person in camera space, mm (x right, y down, z forward): COCO-17 order
_TEMPLATE17 = [(0, -1650, 0), (30, -1680, -20), (-30, -1680, -20), (70, -1660, 40), (-70, -1660, 40),
               (180, -1400, 0), (-180, -1400, 0), (250, -1100, 20), (-250, -1100, 20),
               (270, -850, -30), (-270, -850, -30), (110, -900, 0), (-110, -900, 0),
               (120, -480, 10), (-120, -480, 10), (125, -60, 0), (-125, -60, 0)]

# name -> (n_images, max poses per image, num_aug, n_joints, seed)
FILTER_CASES = {
    'coco17_aug5': (3, 7, 5, 17, 41),
    'coco17_aug1': (2, 5, 1, 17, 42),
    'chain40_aug3': (2, 6, 3, 40, 43),
    'crowd_aug2': (1, 40, 2, 17, 44),
}


def filter_case(name):
    """Per image: distinct people, near-duplicates of some of them (same person detected twice:
    must be suppressed by the pose NMS), and poses that must fail each plausibility test (a bone
    5x too long, augmentation results that disagree, a pose far outside its detection box).
    Decisions are placed well away from the thresholds, so float summation order cannot flip them."""
    n_images, max_poses, A, J, seed = FILTER_CASES[name]
    g = gen(seed)
    if J == 17:
        template = torch.tensor(_TEMPLATE17, dtype=torch.float32)
        edges = list(COCO17_EDGES)
    else:  # a chain skeleton with J joints
        template = torch.cumsum(torch.randn(J, 3, generator=g) * torch.tensor([60.0, 90.0, 40.0]), dim=0)
        edges = [(i, i + 1) for i in range(J - 1)]
    tj = torch.tensor(edges)
    mean_bones = torch.norm(template[tj[:, 0]] - template[tj[:, 1]], dim=-1) * \
        (0.9 + 0.2 * torch.rand(len(edges), generator=g))
    f = 1000.0
    boxes, poses3d, poses2d, kinds = [], [], [], []
    for i in range(n_images):
        n_people = int(torch.randint(1, max(2, max_poses // 2) + 1, (1,), generator=g)) if i != 1 else 0
        items = []  # (pose [A,J,3], score, kind)
        for _ in range(n_people):
            base = template * (0.85 + 0.3 * torch.rand(1, generator=g)) + \
                torch.tensor([0.0, 900.0, 0.0]) + \
                (torch.rand(3, generator=g) - 0.5) * torch.tensor([3000.0, 300.0, 0.0]) + \
                torch.tensor([0.0, 0.0, 3000.0 + 3000.0 * float(torch.rand(1, generator=g))])
            pose = base[None] + torch.randn(A, J, 3, generator=g) * 8
            items.append((pose, 0.5 + 0.5 * float(torch.rand(1, generator=g)), 'person'))
            r = float(torch.rand(1, generator=g))
            if r < 0.4:  # the same person detected twice
                items.append((base[None] + torch.randn(A, J, 3, generator=g) * 12,
                              0.3 + 0.2 * float(torch.rand(1, generator=g)), 'duplicate'))
            elif r < 0.55:  # one bone far too long
                bad = pose.clone()
                bad[:, edges[3][0]] += torch.tensor([0.0, -2500.0, 0.0])
                items.append((bad + torch.tensor([900.0, 0.0, 400.0]), 0.6, 'long_bone'))
            elif r < 0.7 and A > 1:  # augmentation results disagree
                items.append((base[None] + torch.tensor([1500.0, 0.0, 800.0]) +
                              torch.randn(A, J, 3, generator=g) * 900, 0.55, 'inconsistent'))
            elif r < 0.85:  # plausible pose, but nowhere near its detection box
                items.append((pose + torch.tensor([-1200.0, 0.0, 300.0]), 0.7, 'off_box'))
        if i == 0 and items:  # every case holds at least one stretched skeleton
            bad = items[0][0].clone()
            bad[:, edges[3][0]] += torch.tensor([0.0, -2500.0, 0.0])
            items.insert(1, (bad + torch.tensor([900.0, 0.0, 400.0]), 0.6, 'long_bone'))
        items = items[:max_poses]
        order = torch.randperm(len(items), generator=g).tolist()
        items = [items[k] for k in order]
        if items:
            p3 = torch.stack([it[0] for it in items])
            p2 = f * p3[..., :2] / p3[..., 2:] + torch.tensor([960.0, 540.0])
            m2 = p2.mean(dim=1)
            lo, hi = m2.min(dim=1).values, m2.max(dim=1).values
            bx = torch.cat([lo - 20, hi - lo + 40, torch.tensor([[it[1]] for it in items])], dim=1)
            for k, it in enumerate(items):
                if it[2] == 'off_box':
                    bx[k, 0] += 3 * bx[k, 2]
        else:
            p3, p2, bx = torch.zeros(0, A, J, 3), torch.zeros(0, A, J, 2), torch.zeros(0, 5)
        boxes.append(bx.float())
        poses3d.append(p3.float())
        poses2d.append(p2.float())
        kinds.append([it[2] for it in items])
    return dict(boxes=boxes, poses3d=poses3d, poses2d=poses2d, edges=edges, mean_bones=mean_bones.float(),
                n_joints=J, kinds=kinds)


# ------------------------------------------------------------------ checkpoint format (row f.4)

def deterministic_state(state_dict, seed=0):
    """Fills a state_dict with values that depend only on each entry's NAME and shape (so two
    implementations with the same keys get the same weights regardless of construction order):
    conv weights ~ N(0, 1/fan_in), BN scale 0.8..1.2, BN var 0.5..1.5, everything else small."""
    import hashlib
    out = {}
    for k, v in state_dict.items():
        if v.dtype == torch.int64:  # num_batches_tracked
            out[k] = torch.zeros_like(v)
            continue
        h = int(hashlib.sha256(f'{seed}:{k}'.encode()).hexdigest()[:8], 16)
        g = torch.Generator().manual_seed(h)
        if k.endswith('running_var'):
            out[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith('running_mean'):
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif v.dim() == 4:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            out[k] = torch.randn(v.shape, generator=g) * (1.0 / fan_in) ** 0.5
        elif k.endswith('.1.weight') or k.endswith('bn.weight'):
            out[k] = 0.8 + 0.4 * torch.rand(v.shape, generator=g)
        else:
            out[k] = 0.05 * torch.randn(v.shape, generator=g)
    return out


def backbone_probe_input():
    return torch.rand(2, 3, 96, 96, generator=gen(314))


def head_weights_as_consumed(w, feat_dtype):
    """The conv_final weights as the head's GEMM consumes them, for building the expected value on
    the CPU: with 16-bit features the weights are rounded to the feature dtype (what autocast does
    to conv_final in the reference's GPU path; the f16 / bf16 MFMA kernel and
    the library-GEMM path both do that)."""
    if feat_dtype == torch.float32:
        return w
    return w.to(feat_dtype).float()
