"""What arithmetic does the library 1x1 conv use for f32 inputs on this stack?  Error of the logits
vs an fp64 evaluation, for the library conv (MIOpen / rocBLAS through torch) and for torch.matmul.
Developer probe."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
torch.manual_seed(0)
B, C, N, H = 64, 1280, 153, 8
feat = torch.randn(B, C, H, H, device='cuda')
w = torch.randn(N, C, 1, 1, device='cuda') * 0.03
ref = torch.nn.functional.conv2d(feat.double(), w.double())
print('allow_tf32 conv:', torch.backends.cudnn.allow_tf32, 'matmul:', torch.backends.cuda.matmul.allow_tf32)
for name, fn in [('conv2d f32', lambda: torch.nn.functional.conv2d(feat, w)),
                 ('matmul f32', lambda: torch.einsum('nc,bchw->bnhw', w[:, :, 0, 0], feat)),
                 ('conv2d f32 (cudnn.allow_tf32=False)', None)]:
    if fn is None:
        torch.backends.cudnn.allow_tf32 = False
        fn = lambda: torch.nn.functional.conv2d(feat, w)
    y = fn()
    err = (y.double() - ref).abs()
    print(f'{name}: max abs err {float(err.max()):.3e}  rms {float(err.pow(2).mean().sqrt()):.3e}  (|logit| rms {float(ref.pow(2).mean().sqrt()):.3f})')
# an exact-f32 sequential reference for scale: fp32 products summed in fp32 by torch on the CPU
y_cpu = torch.nn.functional.conv2d(feat.cpu(), w.cpu())
err = (y_cpu.double() - ref.cpu()).abs()
print(f'CPU oneDNN f32: max abs err {float(err.max()):.3e} rms {float(err.pow(2).mean().sqrt()):.3e}')
