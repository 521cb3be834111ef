"""Does the explicit ZeroPad2d of the bottom-right stride-2 stage cost anything under autocast +
channels_last?  (developer probe)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd import backbones
x = torch.rand(64, 3, 256, 256, device='cuda')
for centered in (True, False):
    for cl in (True, False):
        net = backbones.efficientnetv2('s', centered_stride=centered).cuda().eval()
        xx = x
        if cl:
            net = net.to(memory_format=torch.channels_last)
            xx = x.contiguous(memory_format=torch.channels_last)
        for dt in (torch.float16, None):
            with torch.inference_mode(), torch.autocast('cuda', dtype=dt or torch.float16, enabled=dt is not None):
                for _ in range(3):
                    y = net(xx)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    y = net(xx)
                torch.cuda.synchronize()
            print(f'centered={centered} channels_last={cl} dtype={dt}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms '
                  f'out channels_last={y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous()}')
