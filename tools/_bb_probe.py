import sys, time, copy, torch
sys.path.insert(0, '/root/repo')
from metrabs_amd.backbones import build_backbone
from bench import calibrate_batchnorm
dev = 'cuda'
torch.manual_seed(0)

def fold_bn(net):
    import torch.nn as nn
    def fuse_seq(seq):
        mods = list(seq.named_children())
        i = 0
        while i < len(mods) - 1:
            (n1, m1), (n2, m2) = mods[i], mods[i + 1]
            if isinstance(m1, nn.Conv2d) and isinstance(m2, nn.BatchNorm2d):
                fused = torch.nn.utils.fusion.fuse_conv_bn_eval(m1, m2)
                setattr(seq, n1, fused); setattr(seq, n2, nn.Identity())
                i += 2
            else:
                i += 1
    for m in net.modules():
        if isinstance(m, torch.nn.Sequential):
            fuse_seq(m)
    return net

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / n * 1e3

net = build_backbone('effnetv2-s').to(dev)
calibrate_batchnorm(net, 256, dev)
net.eval()
x = torch.rand(64, 3, 256, 256, device=dev)
with torch.inference_mode():
    print('fp32 nchw          ms', round(t(lambda: net(x)), 2))
    torch.backends.cudnn.benchmark = True
    print('fp32 nchw bench    ms', round(t(lambda: net(x)), 2))
    netf = fold_bn(copy.deepcopy(net))
    print('fp32 nchw folded   ms', round(t(lambda: netf(x)), 2), 'maxdiff', float((netf(x) - net(x)).abs().max()))
    ncl = copy.deepcopy(netf).to(memory_format=torch.channels_last); xcl = x.contiguous(memory_format=torch.channels_last)
    print('fp32 nhwc folded   ms', round(t(lambda: ncl(xcl)), 2))
    for dt in (torch.float16, torch.bfloat16):
        with torch.autocast('cuda', dtype=dt):
            print(dt, 'nchw folded ms', round(t(lambda: netf(x)), 2))
            print(dt, 'nhwc folded ms', round(t(lambda: ncl(xcl)), 2))
    nh = copy.deepcopy(netf).half().to(memory_format=torch.channels_last); xh = xcl.half()
    print('pure fp16 nhwc folded ms', round(t(lambda: nh(xh)), 2))
    nh2 = copy.deepcopy(netf).half(); xh2 = x.half()
    print('pure fp16 nchw folded ms', round(t(lambda: nh2(xh2)), 2))
