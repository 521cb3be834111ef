"""Two ranks sharing cuda:0 over gloo: where does the time go? (developer probe)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.distributed as dist
from metrabs_amd import distributed
torch.cuda.set_device(0)
rank, world, _ = distributed.init_from_env(backend='gloo')
x = torch.randn(4096, 4096, device='cuda')
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(rank, 'matmul ms', round(t(lambda: x @ x), 2), flush=True)
poses = torch.randn(64, 17, 3, device='cuda'); gathered = torch.empty(world * 64, 17, 3, device='cuda')
print(rank, 'all_gather(list) cuda ms', round(t(lambda: dist.all_gather(list(gathered.chunk(world)), poses)), 2), flush=True)
pc = poses.cpu(); gc = gathered.cpu()
print(rank, 'all_gather(list) cpu ms', round(t(lambda: dist.all_gather(list(gc.chunk(world)), pc)), 2), flush=True)
print(rank, 'all_gather_into_tensor cpu ms', round(t(lambda: dist.all_gather_into_tensor(gc, pc)), 2), flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    y = x @ x; s.synchronize()
    with torch.cuda.graph(g, stream=s):
        y = x @ x
print(rank, 'graph replay ms', round(t(lambda: g.replay()), 2), flush=True)
dist.barrier(); dist.destroy_process_group()
