"""Does TorchScript's tensor-expression fuser (hiprtc) fuse bias-add + SiLU on this ROCm build, and
is the fused kernel faster than the two eager kernels?  Developer probe for the backbone's
elementwise tail (11 % bias adds + 10 % SiLU of the folded f32 backbone)."""
import time, torch

torch._C._jit_set_texpr_fuser_enabled(True)
torch._C._jit_override_can_fuse_on_gpu(True)


@torch.jit.script
def bias_silu(y, b):
    return torch.nn.functional.silu(y + b)


def eager(y, b):
    return torch.nn.functional.silu(y + b)


y = torch.randn(64, 96, 64, 64, device='cuda')
b = torch.randn(1, 96, 1, 1, device='cuda')
with torch.inference_mode():
    for _ in range(5):
        o = bias_silu(y, b)
    torch.cuda.synchronize()
    print('graph:', str(bias_silu.graph_for(y, b))[:600])
    for name, fn in (('eager', eager), ('scripted', bias_silu)):
        for _ in range(3):
            fn(y, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn(y, b)
        torch.cuda.synchronize()
        print(name, (time.perf_counter() - t0) / 50 * 1e6, 'us')
    print('max diff', float((bias_silu(y, b) - eager(y, b)).abs().max()))
