"""Plain-torch.nn restatements of the reference's backbones, random-initialised, for the synthetic
benchmark workloads of BASELINE.json (torchvision is not in this image and there is no network for
checkpoints).  The backbone is NOT part of the hand-written hot path: it stays on PyTorch-ROCm
(MIOpen / rocBLAS), as the north star prescribes.  Only output shape matters to the path:
[B, C, P/32, P/32] with C = 1280 (EffNetV2-S/L, MobileNetV3-L) or 512 (ResNet-18).

Architecture tables follow metrabs_pytorch/backbones/efficientnet.py:379-435 (EfficientNetV2 S/L:
FusedMBConv/MBConv stages, SE ratio 0.25, BN eps 1e-3, PreprocLayer x*2-1 :1181-1186),
metrabs_tf/backbones/resnet.py:746-754 (ResNet-18) and metrabs_tf/backbones/mobilenet_v3.py:387-432
(MobileNetV3-Large, last_point_ch 1280).  Padding is symmetric k//2 (the reference uses TF-'SAME'
fixed padding, efficientnet.py:1127-1161; irrelevant for throughput).
"""
import collections
import threading

import torch
import torch.nn.functional as F
from torch import nn


class DepthwiseConv2d(nn.Conv2d):
    """A depthwise convolution that runs on PyTorch's own HIP depthwise kernel instead of MIOpen.
    MIOpen serves f32 / f16 NCHW depthwise 3x3 with its naive direct kernel on this stack; with
    MIOpen switched off for these layers the EfficientNetV2-S forward at the bench shape takes
    13.13 instead of 14.02 ms in f32 (10.87 vs 11.71 ms under f16 autocast), features equal to
    5e-6 relative (tools/experiments/depthwise_backend_probe.py).  Same parameters, same
    state_dict keys; still PyTorch-ROCm, only the backend choice of these layers changes."""

    use_miopen = False  # class-wide switch (tools/experiments/depthwise_backend_probe.py flips it)
    _backend_lock = threading.Lock()  # the MIOpen switch is process-global state in torch

    def forward(self, x):
        if x.is_cuda and not DepthwiseConv2d.use_miopen and torch.backends.cudnn.enabled:
            # scoped; the lock serialises THESE layers among themselves (two of them must not
            # interleave their save / restore of the flag).  It does not isolate other threads'
            # convolutions: a convolution another thread launches while this context is open sees
            # MIOpen off too (the switch is process-global in torch).  Every other flag is passed
            # through unchanged -- flags() would otherwise reset benchmark / deterministic /
            # allow_tf32 to its keyword defaults inside the context.
            cudnn = torch.backends.cudnn
            with DepthwiseConv2d._backend_lock, cudnn.flags(
                    enabled=False, benchmark=cudnn.benchmark, deterministic=cudnn.deterministic,
                    allow_tf32=cudnn.allow_tf32):
                return super().forward(x)
        return super().forward(x)


class ConvBNAct(nn.Sequential):
    """conv '0' + batch norm '1' (+ activation '2'): the parameter names of torchvision's
    Conv2dNormActivation, which the reference's checkpoints use."""

    def __init__(self, cin, cout, k=3, s=1, groups=1, act=nn.SiLU, eps=1e-3, padding=None):
        conv = DepthwiseConv2d if groups == cin == cout and groups > 1 else nn.Conv2d
        layers = [conv(cin, cout, k, s, k // 2 if padding is None else padding, groups=groups,
                       bias=False),
                  nn.BatchNorm2d(cout, eps=eps)]
        if act is not None:
            layers.append(act())
        super().__init__(*layers)


class SqueezeExcite(nn.Module):
    def __init__(self, channels, squeeze, gate=nn.Sigmoid, act=nn.SiLU):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, channels, 1)
        self.act, self.gate = act(), gate()

    # set by fold_batchnorm(fused_epilogue=True): the ConvBiasAct in front of this block, whose
    # epilogue pass also produced the per-channel mean of the tensor it handed over
    mean_from = ()

    def forward(self, x):
        s = None
        for src in self.mean_from:
            s = src.take_mean(x)
        if s is None:
            s = x.mean((2, 3), keepdim=True)
        return x * self.gate(self.fc2(self.act(self.fc1(s))))


def _padded_conv(layers, name, cin, cout, k, stride, groups=1, act=nn.SiLU, bottomright=False):
    """The reference pads explicitly (TF 'SAME' independent of the input size,
    efficientnet.py:1127-1161: (k-1)//2 before, the rest after, shifted one pixel to the bottom
    right for the `bottomright_stride` layer) and convolves unpadded.  Symmetric padding is folded
    into the convolution (same zeros, one op less); the asymmetric case keeps a ZeroPad2d, which
    has no parameters, so the checkpoint keys are unaffected either way."""
    total = k - 1
    beg, end = total // 2, total - total // 2
    if bottomright:
        layers['padding'] = nn.ZeroPad2d((beg - 1, end + 1, beg - 1, end + 1))
        layers[name] = ConvBNAct(cin, cout, k, stride, groups=groups, act=act, padding=0)
    elif beg == end:
        layers[name] = ConvBNAct(cin, cout, k, stride, groups=groups, act=act, padding=beg)
    else:
        layers['padding'] = nn.ZeroPad2d((beg, end, beg, end))
        layers[name] = ConvBNAct(cin, cout, k, stride, groups=groups, act=act, padding=0)


class FusedMBConv(nn.Module):
    """efficientnet.py:176-234 (module names '0' [, '1'] inside `block`)."""

    def __init__(self, cin, cout, expand, stride, bottomright=False):
        super().__init__()
        self.residual = stride == 1 and cin == cout
        mid = cin * expand
        layers = collections.OrderedDict()
        if expand == 1:
            _padded_conv(layers, '0', cin, cout, 3, stride, bottomright=bottomright)
        else:
            _padded_conv(layers, '0', cin, mid, 3, stride, bottomright=bottomright)
            layers['1'] = ConvBNAct(mid, cout, 1, 1, act=None)
        self.block = nn.Sequential(layers)

    def forward(self, x):
        return _block_plus_skip(self.block, x) if self.residual else self.block(x)


class MBConv(nn.Module):
    """efficientnet.py:110-173: '0' expand 1x1, '1' depthwise, '2' squeeze-excite (fc1, fc2),
    '3' project."""

    def __init__(self, cin, cout, expand, stride, k=3, bottomright=False):
        super().__init__()
        self.residual = stride == 1 and cin == cout
        mid = cin * expand
        layers = collections.OrderedDict()
        if mid != cin:
            layers['0'] = ConvBNAct(cin, mid, 1, 1)
        n = len(layers)
        _padded_conv(layers, str(n), mid, mid, k, stride, groups=mid, bottomright=bottomright)
        layers[str(n + 1)] = SqueezeExcite(mid, max(1, cin // 4))
        layers[str(n + 2)] = ConvBNAct(mid, cout, 1, 1, act=None)
        self.block = nn.Sequential(layers)

    def forward(self, x):
        return _block_plus_skip(self.block, x) if self.residual else self.block(x)


class Preproc(nn.Module):
    """PreprocLayer (efficientnet.py:1181-1186): [0,1] -> [-1,1]."""

    def forward(self, x):
        return x * 2 - 1


EFFNETV2 = {
    # (block, expand, stride, cin, cout, n_layers); efficientnet.py:399-431.  The LAST stride-2 stage
    # is the `bottomright_stride=FLAGS.centered_stride` one.
    's': dict(stem=24, stages=[('f', 1, 1, 24, 24, 2), ('f', 4, 2, 24, 48, 4), ('f', 4, 2, 48, 64, 4),
                               ('m', 4, 2, 64, 128, 6), ('m', 6, 1, 128, 160, 9),
                               ('m', 6, 2, 160, 256, 15)], head=1280),
    'm': dict(stem=24, stages=[('f', 1, 1, 24, 24, 3), ('f', 4, 2, 24, 48, 5), ('f', 4, 2, 48, 80, 5),
                               ('m', 4, 2, 80, 160, 7), ('m', 6, 1, 160, 176, 14),
                               ('m', 6, 2, 176, 304, 18), ('m', 6, 1, 304, 512, 5)], head=1280),
    'l': dict(stem=32, stages=[('f', 1, 1, 32, 32, 4), ('f', 4, 2, 32, 64, 7), ('f', 4, 2, 64, 96, 7),
                               ('m', 4, 2, 96, 192, 10), ('m', 6, 1, 192, 224, 19),
                               ('m', 6, 2, 224, 384, 25), ('m', 6, 1, 384, 640, 7)], head=1280),
}


def efficientnetv2(size='s', centered_stride=True):
    """`Sequential(PreprocLayer(), efficientnet_v2_<size>().features)` of the reference
    (scripts/demo_image.py:63-66) with the same module tree, hence the same state_dict keys
    ('1.0.0.weight' = stem conv, '1.<stage>.<i>.block.<k>....', '1.<last>.0.weight' = 1x1 head conv)
    and the same arithmetic (checked against the reference's own class on shared weights,
    tests/test_oracle_pin.py)."""
    cfg = EFFNETV2[size]
    feats = collections.OrderedDict()
    _padded_conv(feats, '0', 3, cfg['stem'], 3, 2)
    last_s2 = max(i for i, st in enumerate(cfg['stages']) if st[2] == 2)
    for si, (kind, expand, stride, cin, cout, n) in enumerate(cfg['stages']):
        blk = FusedMBConv if kind == 'f' else MBConv
        stage = [blk(cin if i == 0 else cout, cout, expand, stride if i == 0 else 1,
                     bottomright=(centered_stride and si == last_s2 and i == 0)) for i in range(n)]
        feats[str(si + 1)] = nn.Sequential(*stage)
    feats[str(len(cfg['stages']) + 1)] = ConvBNAct(cfg['stages'][-1][4], cfg['head'], 1, 1)
    net = nn.Sequential(Preproc(), nn.Sequential(feats))
    net.out_channels = cfg['head']
    return net


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.a = ConvBNAct(cin, cout, 3, stride, act=nn.ReLU, eps=1e-5)
        self.b = ConvBNAct(cout, cout, 3, 1, act=None, eps=1e-5)
        self.down = None if stride == 1 and cin == cout else ConvBNAct(cin, cout, 1, stride, act=None, eps=1e-5)

    def forward(self, x):
        idt = x if self.down is None else self.down(x)
        return torch.relu(self.b(self.a(x)) + idt)


def resnet18():
    layers = [ConvBNAct(3, 64, 7, 2, act=nn.ReLU, eps=1e-5), nn.MaxPool2d(3, 2, 1)]
    cin = 64
    for cout, stride in [(64, 1), (128, 2), (256, 2), (512, 2)]:
        layers += [BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1)]
        cin = cout
    net = nn.Sequential(*layers)
    net.out_channels = 512
    return net


class MBv3Block(nn.Module):
    def __init__(self, cin, k, exp, cout, se, hs, stride):
        super().__init__()
        act = nn.Hardswish if hs else nn.ReLU
        self.residual = stride == 1 and cin == cout
        layers = []
        if exp != cin:
            layers.append(ConvBNAct(cin, exp, 1, 1, act=act))
        layers.append(ConvBNAct(exp, exp, k, stride, groups=exp, act=act))
        if se:
            layers.append(SqueezeExcite(exp, max(8, exp // 4), gate=nn.Hardsigmoid, act=nn.ReLU))
        layers.append(ConvBNAct(exp, cout, 1, 1, act=None))
        self.block = nn.Sequential(*layers)

    def forward(self, x):
        return _block_plus_skip(self.block, x) if self.residual else self.block(x)


def mobilenet_v3_large():
    table = [(3, 16, 16, 0, 0, 1), (3, 64, 24, 0, 0, 2), (3, 72, 24, 0, 0, 1), (5, 72, 40, 1, 0, 2),
             (5, 120, 40, 1, 0, 1), (5, 120, 40, 1, 0, 1), (3, 240, 80, 0, 1, 2), (3, 200, 80, 0, 1, 1),
             (3, 184, 80, 0, 1, 1), (3, 184, 80, 0, 1, 1), (3, 480, 112, 1, 1, 1), (3, 672, 112, 1, 1, 1),
             (5, 672, 160, 1, 1, 2), (5, 960, 160, 1, 1, 1), (5, 960, 160, 1, 1, 1)]
    layers = [Preproc(), ConvBNAct(3, 16, 3, 2, act=nn.Hardswish)]
    cin = 16
    for k, exp, cout, se, hs, s in table:
        layers.append(MBv3Block(cin, k, exp, cout, bool(se), bool(hs), s))
        cin = cout
    layers += [ConvBNAct(cin, 960, 1, 1, act=nn.Hardswish), ConvBNAct(960, 1280, 1, 1, act=nn.Hardswish)]
    net = nn.Sequential(*layers)
    net.out_channels = 1280
    return net


def build_backbone(name):
    name = name.lower()
    if name in ('efficientnetv2-s', 'effnetv2-s', 'effv2s'):
        return efficientnetv2('s')
    if name in ('efficientnetv2-l', 'effnetv2-l', 'effv2l'):
        return efficientnetv2('l')
    if name in ('resnet18', 'resnet-18'):
        return resnet18()
    if name in ('mobilenetv3', 'mobilenetv3-large', 'mobilenet-v3'):
        return mobilenet_v3_large()
    raise ValueError(f'unknown backbone {name}')


_ACT_NAMES = {nn.SiLU: 'silu', nn.ReLU: 'relu', nn.Hardswish: 'hardswish'}


class ConvBiasAct(nn.Module):
    """A folded conv + BN (+ activation): the convolution without its bias (MIOpen / rocBLAS), then
    "+ bias[c]" and the activation as ONE in-place pass (kernels.bias_act_, K10) instead of the two
    elementwise kernels PyTorch-ROCm would launch.  CPU tensors take the plain torch ops."""

    def __init__(self, conv, bias, act):
        super().__init__()
        self.conv = conv
        self.register_buffer('bias', bias.detach().float().contiguous())
        self.act = act
        self.act_name = None if act is None else _ACT_NAMES[type(act)]
        self.emit_mean = False  # a squeeze-excite block follows: give it its x.mean((2, 3)) for free
        self._mean = None

    def take_mean(self, x):
        """The [B, C, 1, 1] mean of `x` if `x` is the very tensor this module returned last."""
        held, self._mean = self._mean, None
        if held is not None and held[0] is x:
            return held[1].to(x.dtype).view(x.shape[0], x.shape[1], 1, 1)
        return None

    def forward(self, x, residual=None):
        y = self.conv(x)
        # K10 moves 16 bytes per lane: planes of a multiple of the vector width (7x7 maps at 224 px,
        # 5x5 at 160 px are not), 16-byte aligned storage; everything else takes the torch ops
        hw_vec_ok = (y.shape[2] * y.shape[3]) % (16 // y.element_size()) == 0
        if y.is_cuda and y.is_contiguous() and hw_vec_ok and y.data_ptr() % 16 == 0 and (
                residual is None or (residual.dtype == y.dtype and residual.is_contiguous()
                                     and residual.data_ptr() % 16 == 0)):
            from . import kernels
            if self.emit_mean and residual is None:
                y, mean = kernels.bias_act_rowmean_(y, self.bias, self.act_name)
                self._mean = (y, mean)
                return y
            return kernels.bias_act_(y, self.bias, self.act_name, residual)
        y = y + self.bias.view(1, -1, 1, 1).to(y.dtype)
        y = y if self.act is None else self.act(y)
        return y if residual is None else residual + y


class DepthwiseBiasAct(nn.Module):
    """A folded depthwise 3x3 conv + BN + activation as ONE HIP pass over the plane (K11): the
    convolution, "+ bias", the activation and -- in front of a squeeze-excite block -- the
    per-channel mean, instead of PyTorch's depthwise kernel followed by K10.  Other tensors (CPU,
    non-contiguous, widths that are not a multiple of 4) take the torch ops."""

    def __init__(self, conv, bias, act):
        super().__init__()
        self.weight = nn.Parameter(conv.weight.detach().float().contiguous(), requires_grad=False)
        self.register_buffer('bias', bias.detach().float().contiguous())
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        # (left, right, top, bottom) of a ZeroPad2d that stood in front of the layer and was folded
        # in by fold_batchnorm (the TF-'SAME' padding of the stride-2 layers); None: self.pad all round
        self.pads = None
        self.act = act
        self.act_name = None if act is None else _ACT_NAMES[type(act)]
        self.emit_mean = False
        self._mean = None

    @staticmethod
    def applies_to(conv):
        return (isinstance(conv, nn.Conv2d) and conv.groups == conv.in_channels == conv.out_channels
                and conv.groups > 1 and conv.kernel_size == (3, 3) and conv.dilation == (1, 1)
                and conv.stride in ((1, 1), (2, 2)) and conv.padding in ((0, 0), (1, 1))
                and conv.padding_mode == 'zeros')

    take_mean = ConvBiasAct.take_mean

    def forward(self, x):
        pl, pr = (self.pad, self.pad) if self.pads is None else self.pads[:2]
        ow = (x.shape[3] + pl + pr - 3) // self.stride + 1
        pad = self.pad if self.pads is None else self.pads
        if x.is_cuda and x.is_contiguous() and ow % 4 == 0 and \
                x.dtype in (torch.float32, torch.float16, torch.bfloat16):
            from . import kernels
            if self.emit_mean:
                y, mean = kernels.depthwise3x3_bias_act(x, self.weight, self.bias, self.act_name,
                                                        self.stride, pad, want_mean=True)
                self._mean = (y, mean)
                return y
            return kernels.depthwise3x3_bias_act(x, self.weight, self.bias, self.act_name,
                                                 self.stride, pad)
        if self.pads is not None:
            x = F.pad(x, self.pads)
        y = F.conv2d(x, self.weight.to(x.dtype), self.bias.to(x.dtype), self.stride,
                     self.pad if self.pads is None else 0, groups=self.weight.shape[0])
        return y if self.act is None else self.act(y)


def _block_plus_skip(block, x):
    """x + block(x) of an (Fused)MBConv; when the block ends in a folded convolution (ConvBiasAct)
    the skip connection rides on its epilogue instead of being a kernel of its own."""
    last = block[-1]
    tail = last[0] if isinstance(last, ConvBNAct) else None
    if isinstance(tail, ConvBiasAct):
        y = x
        for m in list(block)[:-1]:
            y = m(y)
        for m in list(last)[1:]:
            assert isinstance(m, nn.Identity)
        return tail(y, residual=x)
    return x + block(x)


def fold_batchnorm(backbone, fused_epilogue=False):
    """Inference-time copy of `backbone` with every batch norm folded into the convolution in front
    of it (w' = w * gamma / sqrt(var + eps), b' = beta - mean * gamma / sqrt(var + eps)): the same
    function up to rounding (features equal to ~1e-5 relative in f32), one elementwise pass over
    every activation less.  EfficientNetV2-S forward at the bench shape: 13.3 -> 11.8 ms in f32,
    11.6 -> 10.2 ms under f16 autocast (tools/experiments/bn_backend_probe.py).  The original keeps
    its checkpoint-compatible parameters; the copy has conv biases and no BatchNorm2d.
    fused_epilogue=True additionally runs "+ bias, activation" behind each folded convolution as one
    in-place HIP pass (ConvBiasAct) instead of PyTorch-ROCm's two elementwise kernels."""
    import copy
    from torch.nn.utils.fusion import fuse_conv_bn_eval
    if backbone.training:
        raise ValueError('fold_batchnorm needs the running statistics of an eval-mode network')
    folded = copy.deepcopy(backbone)
    for m in folded.modules():
        if isinstance(m, ConvBNAct) and isinstance(m[1], nn.BatchNorm2d):
            conv = fuse_conv_bn_eval(m[0], m[1])
            act = m[2] if len(m) > 2 else None
            if fused_epilogue and (act is None or type(act) in _ACT_NAMES):
                bias, conv.bias = conv.bias, None
                m[0] = (DepthwiseBiasAct if DepthwiseBiasAct.applies_to(conv) else ConvBiasAct)(
                    conv, bias, act)
                if act is not None:
                    m[2] = nn.Identity()
            else:
                m[0] = conv
            m[1] = nn.Identity()
    if any(isinstance(m, nn.BatchNorm2d) for m in folded.modules()):
        raise ValueError('a BatchNorm2d outside a ConvBNAct block cannot be folded here')
    if fused_epilogue:  # ZeroPad2d -> depthwise 3x3: the padding becomes an argument of K11
        for seq in folded.modules():
            if not isinstance(seq, nn.Sequential) or isinstance(seq, ConvBNAct):
                continue
            names = list(seq._modules)
            for n_pad, n_conv in zip(names, names[1:]):
                pad, blk = seq._modules[n_pad], seq._modules[n_conv]
                if isinstance(pad, nn.ZeroPad2d) and isinstance(blk, ConvBNAct) and \
                        isinstance(blk[0], DepthwiseBiasAct) and blk[0].pad == 0:
                    l, r, t, b = (int(p) for p in pad.padding)
                    if 0 <= l <= 1 and 0 <= t <= 1 and 0 <= r <= 2 and 0 <= b <= 2:
                        blk[0].pads = (l, r, t, b)
                        seq._modules[n_pad] = nn.Identity()
    if fused_epilogue:  # conv -> squeeze-excite: the epilogue pass also emits the channel means
        for seq in folded.modules():
            if not isinstance(seq, nn.Sequential) or isinstance(seq, ConvBNAct):
                continue
            kids = list(seq)
            for prev, nxt in zip(kids, kids[1:]):
                if isinstance(nxt, SqueezeExcite) and isinstance(prev, ConvBNAct) and \
                        isinstance(prev[0], (ConvBiasAct, DepthwiseBiasAct)) and \
                        all(isinstance(m, nn.Identity) for m in list(prev)[1:]):
                    prev[0].emit_mean = True
                    nxt.mean_from = (prev[0],)
    return folded


def calibrate_batchnorm(backbone, res, dev, batches=2, batch_size=16, seed=7, samples=None):
    """Random-weight networks with untouched BatchNorm statistics (mean 0 / var 1) let activations
    grow layer by layer until they overflow.  A few forward passes in training mode on synthetic
    crops set the running statistics (cumulative average), which keeps every layer at unit scale --
    the regime a trained checkpoint is in.  Weights stay random; the FLOPs are unchanged.
    samples: optional f32 crops [n, 3, res, res] of the kind the network will actually see (the
    benchmark's sampler output: low-contrast resampled noise with zero padding, whose statistics
    differ enough from uniform noise for an f16 EfficientNetV2-L to overflow); they are used in
    chunks of batch_size beside the uniform-noise batches."""
    bns = [m for m in backbone.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for m in bns:
        m.momentum = None
        m.reset_running_stats()
    backbone.train()
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for _ in range(batches):
            backbone(torch.rand(batch_size, 3, res, res, device=dev, generator=g))
        if samples is not None:
            for chunk in samples.float().split(batch_size):
                if len(chunk) > 1:
                    backbone(chunk)
    backbone.eval()
    return backbone
