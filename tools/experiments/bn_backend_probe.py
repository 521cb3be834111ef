"""Inference-time batch norm of the (PyTorch-ROCm) backbone: MIOpen's BN kernel vs PyTorch's native
one vs folding BN into the preceding convolution.  EfficientNetV2-S forward at the bench shape."""
import copy, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd.backbones import build_backbone, calibrate_batchnorm, ConvBNAct
from torch.nn.utils.fusion import fuse_conv_bn_eval

torch.manual_seed(0)
net = calibrate_batchnorm(build_backbone('effnetv2-s').cuda(), 256, 'cuda', batch_size=4).eval()
x = torch.rand(64, 3, 256, 256, device='cuda')


def run(model, dtype):
    def fwd():
        if dtype is None:
            return model(x)
        with torch.autocast('cuda', dtype=dtype):
            return model(x)
    with torch.inference_mode():
        for _ in range(3):
            y = fwd()
        torch.cuda.synchronize()
        st = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            fwd(); st.synchronize()
            with torch.cuda.graph(g, stream=st):
                y = fwd()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 10 * 1e3, y.float().clone()


class NativeBN(torch.nn.BatchNorm2d):
    def forward(self, inp):
        torch.backends.cudnn.enabled = False
        try:
            return super().forward(inp)
        finally:
            torch.backends.cudnn.enabled = True


native = copy.deepcopy(net)
for m in native.modules():
    if type(m) is torch.nn.BatchNorm2d:
        m.__class__ = NativeBN
folded = copy.deepcopy(net)
n_fold = 0
for m in folded.modules():
    if isinstance(m, ConvBNAct):
        m[0] = fuse_conv_bn_eval(m[0], m[1])
        m[1] = torch.nn.Identity()
        n_fold += 1
left = sum(isinstance(m, torch.nn.BatchNorm2d) for m in folded.modules())
print(f'folded {n_fold} conv+BN pairs, {left} BatchNorm2d left')
from metrabs_amd.backbones import fold_batchnorm
fused = fold_batchnorm(net, fused_epilogue=True)
for dtype in (None, torch.float16):
    t3, y3 = run(fused, dtype)
    print(f'{dtype}: folded + K10 bias/activation epilogue {t3:.2f} ms', flush=True)
    t0, y0 = run(net, dtype)
    t1, y1 = run(native, dtype)
    t2, y2 = run(folded, dtype)
    r = lambda a, b: float((a - b).abs().max() / a.abs().max())
    print(f'{dtype}: MIOpen BN {t0:.2f} ms | native BN {t1:.2f} ms (rel diff {r(y0, y1):.1e}) | folded {t2:.2f} ms '
          f'(rel diff {r(y0, y2):.1e})', flush=True)
