"""Which backend should the depthwise convolutions of the (PyTorch-ROCm) backbone use?  MIOpen
serves f32 NCHW depthwise 3x3 with its naive kernel (44 % of the bench step's kernel time,
profiles/r01g_kernel_trace_bench_f32.md); with MIOpen disabled for those layers PyTorch runs its
own depthwise kernel.  Times the EfficientNetV2-S forward at the bench shape both ways."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd.backbones import build_backbone, calibrate_batchnorm

name = sys.argv[1] if len(sys.argv) > 1 else 'effnetv2-s'
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
torch.manual_seed(0)
net = calibrate_batchnorm(build_backbone(name).cuda(), res, 'cuda', batch_size=4).eval()
dw = [m for m in net.modules() if isinstance(m, torch.nn.Conv2d) and m.groups == m.in_channels and m.groups > 1]
print(f'{name}: {len(dw)} depthwise convs')
x = torch.rand(B, 3, res, res, device='cuda')


def run(dtype):
    def fwd():
        if dtype is None:
            return net(x)
        with torch.autocast('cuda', dtype=dtype):
            return net(x)
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


from metrabs_amd.backbones import DepthwiseConv2d
assert all(isinstance(m, DepthwiseConv2d) for m in dw)
for dtype in (None, torch.float16):
    DepthwiseConv2d.use_miopen = True
    t_miopen, y0 = run(dtype)
    DepthwiseConv2d.use_miopen = False
    t_native, y1 = run(dtype)
    rel = float((y0 - y1).abs().max() / y0.abs().max())
    print(f'{name} B={B} {res}px {dtype}: MIOpen depthwise {t_miopen:.2f} ms, PyTorch-native depthwise {t_native:.2f} ms, '
          f'max rel diff of the features {rel:.2e}', flush=True)
