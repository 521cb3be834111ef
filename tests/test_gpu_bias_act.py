"""GPU: K10, the in-place bias + activation epilogue of the backbone's inference copy, against the
torch ops it replaces; and the folded + fused backbone against the original network."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {None: lambda t: t, 'relu': F.relu, 'silu': F.silu, 'hardswish': F.hardswish}


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape', [(3, 5, 8, 8), (2, 1536, 8, 8), (64, 24, 128, 128), (1, 7, 2, 4), (5, 960, 16, 16)])
@pytest.mark.parametrize('act', [None, 'relu', 'silu', 'hardswish'])
def test_bias_act_vs_torch(shape, act, dtype, hip_lib):
    from metrabs_amd import kernels
    g = torch.Generator(device='cuda').manual_seed(sum(shape))
    y = (torch.randn(shape, device='cuda', generator=g) * 3).to(dtype)
    b = torch.randn(shape[1], device='cuda', generator=g)
    want = ACTS[act](y.float() + b.view(1, -1, 1, 1))
    got = kernels.bias_act_(y.clone(), b, act)
    assert got.dtype == dtype and got.shape == y.shape
    tol = {torch.float32: 2e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
    err = float(((got.float() - want).abs() / (1 + want.abs())).max())
    assert err <= tol, err
    # in place
    y2 = y.clone()
    assert kernels.bias_act_(y2, b, act).data_ptr() == y2.data_ptr()
    # with the skip connection riding on the same pass: act(y + b) + r
    r = (torch.randn(shape, device='cuda', generator=g) * 2).to(dtype)
    got_r = kernels.bias_act_(y.clone(), b, act, residual=r)
    want_r = want + r.float()
    err_r = float(((got_r.float() - want_r).abs() / (1 + want_r.abs())).max())
    assert err_r <= tol, err_r


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('shape', [(3, 5, 8, 8), (2, 1536, 8, 8), (64, 960, 16, 16), (1, 7, 2, 4), (2, 9, 28, 28),
                                   (130, 3, 4, 4)])
def test_bias_act_rowmean_vs_torch(shape, dtype, hip_lib):
    """The row-mean variant: same result tensor as bias_act_, and the mean over H*W of that result."""
    from metrabs_amd import kernels
    g = torch.Generator(device='cuda').manual_seed(sum(shape) + 1)
    y = (torch.randn(shape, device='cuda', generator=g) * 3).to(dtype)
    b = torch.randn(shape[1], device='cuda', generator=g)
    plain = kernels.bias_act_(y.clone(), b, 'silu')
    got, mean = kernels.bias_act_rowmean_(y.clone(), b, 'silu')
    assert torch.equal(got, plain) and mean.shape == shape[:2] and mean.dtype == torch.float32
    want = plain.float().mean((2, 3))
    assert float((mean - want).abs().max()) <= 1e-5 * (1 + float(want.abs().max()))


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('cfg', [(2, 960, 16, 16, 1, 1), (3, 256, 32, 32, 2, 1), (2, 96, 18, 18, 2, 0),
                                 (5, 1536, 8, 8, 1, 1), (1, 7, 4, 8, 1, 1), (2, 5, 9, 9, 2, 0), (70, 3, 16, 16, 1, 1),
                                 (3, 4, 32, 32, 1, 1), (2, 5, 64, 64, 2, 1), (2, 6, 10, 18, 1, 0),
                                 (64, 960, 16, 16, 1, 1), (33, 130, 8, 8, 1, 1), (2, 3, 24, 40, 1, 1)])
def test_depthwise3x3_bias_act_vs_torch(cfg, dtype, hip_lib):
    """K11 vs F.conv2d(groups=C) + bias + SiLU and the mean of that: every (stride, pad) the backbones
    use, planes of 2 .. 256 output vectors (several 64-group items per plane: the mean spans them),
    plane counts that do not fill the last wave, launches of one item per wave and of many (the
    persistent grid's three-deep prefetch ring and its tails), aligned-vector and scalar rows."""
    from metrabs_amd import kernels
    B, C, H, W, stride, pad = cfg
    g = torch.Generator(device='cuda').manual_seed(sum(cfg))
    x = torch.randn(B, C, H, W, device='cuda', generator=g).to(dtype)
    w = torch.randn(C, 1, 3, 3, device='cuda', generator=g) * 0.4
    b = torch.randn(C, device='cuda', generator=g)
    want = F.silu(F.conv2d(x.float(), w, b, stride, pad, groups=C))
    got, mean = kernels.depthwise3x3_bias_act(x, w, b, 'silu', stride, pad, want_mean=True)
    assert got.shape == want.shape and got.dtype == dtype
    tol = 3e-6 if dtype == torch.float32 else 2e-3
    assert float(((got.float() - want).abs() / (1 + want.abs())).max()) <= tol
    assert float((mean - got.float().mean((2, 3))).abs().max()) <= 1e-5 * (1 + float(want.abs().max()))
    plain = kernels.depthwise3x3_bias_act(x, w, b, 'silu', stride, pad)
    assert torch.equal(plain, got)
    none = kernels.depthwise3x3_bias_act(x, w, b, None, stride, pad)
    want_none = F.conv2d(x.float(), w, b, stride, pad, groups=C)
    assert float(((none.float() - want_none).abs() / (1 + want_none.abs())).max()) <= tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('cfg', [(3, 960, 16, 16, 2, (0, 2, 0, 2)), (2, 256, 32, 32, 2, (0, 1, 0, 1)),
                                 (2, 7, 8, 24, 2, (1, 0, 1, 0)), (2, 5, 16, 16, 1, (1, 1, 1, 1)),
                                 (2, 6, 14, 12, 1, (0, 2, 1, 1)), (70, 3, 16, 16, 2, (0, 2, 0, 2))])
def test_depthwise3x3_with_folded_zero_padding(cfg, dtype, hip_lib):
    """The explicit ZeroPad2d of the reference's TF-'SAME' stride-2 layers (efficientnet.py:1127-1161:
    (0,1,0,1); (0,2,0,2) for the bottomright_stride layer) as an argument of K11: vs F.pad + F.conv2d
    on the padded copy.  Aligned-vector rows with the edge element on the right (left padding 0), on
    the left (left padding 1), and the scalar path (stride 1 with asymmetric padding)."""
    from metrabs_amd import kernels
    B, C, H, W, stride, pads = cfg
    g = torch.Generator(device='cuda').manual_seed(sum(cfg[:5]) + sum(pads))
    x = torch.randn(B, C, H, W, device='cuda', generator=g).to(dtype)
    w = torch.randn(C, 1, 3, 3, device='cuda', generator=g) * 0.4
    b = torch.randn(C, device='cuda', generator=g)
    want = F.silu(F.conv2d(F.pad(x.float(), pads), w, b, stride, 0, groups=C))
    got, mean = kernels.depthwise3x3_bias_act(x, w, b, 'silu', stride, pads, want_mean=True)
    assert got.shape == want.shape and got.dtype == dtype
    tol = 3e-6 if dtype == torch.float32 else 2e-3
    assert float(((got.float() - want).abs() / (1 + want.abs())).max()) <= tol
    assert float((mean - got.float().mean((2, 3))).abs().max()) <= 1e-5 * (1 + float(want.abs().max()))


def test_fold_batchnorm_folds_the_zero_padding_into_k11(hip_lib):
    from metrabs_amd import backbones
    net = backbones.build_backbone('effnetv2-s').eval()
    fused = backbones.fold_batchnorm(net, fused_epilogue=True)
    padded = [m for m in fused.modules() if isinstance(m, backbones.DepthwiseBiasAct) and m.pads is not None]
    assert [m.pads for m in padded] == [(0, 2, 0, 2)]  # the bottomright_stride layer
    n_pad = lambda n: sum(isinstance(m, torch.nn.ZeroPad2d) for m in n.modules())
    assert n_pad(fused) == n_pad(net) - 1


def test_bias_act_rejects_what_it_cannot_vectorise(hip_lib):
    from metrabs_amd import kernels
    y = torch.zeros(2, 3, 3, 3, device='cuda')  # H*W = 9: a 16-byte vector would straddle channels
    with pytest.raises(RuntimeError):
        kernels.bias_act_(y, torch.zeros(3, device='cuda'), 'silu')
    with pytest.raises(ValueError):
        kernels.bias_act_(torch.zeros(2, 4, 4, 4, device='cuda').permute(0, 2, 3, 1), torch.zeros(4, device='cuda'), None)


@pytest.mark.parametrize('name,res', [('effnetv2-s', 256), ('mobilenetv3', 256), ('resnet18', 256),
                                      ('effnetv2-s', 224), ('effnetv2-s', 160), ('mobilenetv3', 224)])
def test_folded_fused_backbone_is_the_same_function(name, res, hip_lib):
    """(224 / 160 px: 7x7 = 49 and 5x5 = 25-position maps are not a multiple of K10's 16-byte vectors:
    those layers must take the torch ops, not raise.)"""
    from metrabs_amd import backbones
    torch.manual_seed(0)
    net = backbones.calibrate_batchnorm(backbones.build_backbone(name).cuda(), res, 'cuda', batch_size=4)
    fused = backbones.fold_batchnorm(net, fused_epilogue=True)
    assert any(isinstance(m, backbones.DepthwiseBiasAct) for m in fused.modules()) == (name != 'resnet18')
    n_se = sum(isinstance(m, backbones.SqueezeExcite) for m in fused.modules())
    assert sum(bool(m.mean_from) for m in fused.modules() if isinstance(m, backbones.SqueezeExcite)) == n_se
    x = torch.rand(4, 3, res, res, device='cuda')
    with torch.inference_mode():
        a, b = net(x), fused(x)
        with torch.autocast('cuda', dtype=torch.float16):
            c = fused(x)
    assert float((a - b).abs().max()) <= 1e-3 * float(a.abs().max())
    assert c.dtype == torch.float16 and float((a - c.float()).abs().max()) <= 0.1 * float(a.abs().max())
