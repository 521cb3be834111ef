import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.distributed as dist
import bench
from metrabs_amd import distributed
from metrabs_amd.pipeline import GraphedCropPipeline
sys.argv = ['bench.py', '--gpus', '2']
args = bench.parse_args()
torch.cuda.set_device(0)
rank, world, _ = distributed.init_from_env(backend='gloo')
est, cfg = bench.build_model(args, torch.device('cuda', 0))
pipe = GraphedCropPipeline(est, 8, 1080, 1920, 64, num_aug=1)
bench.synth_inputs(pipe, 8, 1080, 1920, 64, seed=100 + rank)
pipe.capture()
gathered = torch.empty(world * 64, 17, 3, device='cuda')
def sync(): torch.cuda.synchronize()
for i in range(4):
    sync(); t0 = time.perf_counter(); poses = pipe.run(); sync(); t1 = time.perf_counter()
    dist.all_gather(list(gathered.chunk(world)), poses.contiguous()); sync(); t2 = time.perf_counter()
    print(rank, i, 'run ms', round((t1 - t0) * 1e3, 2), 'gather ms', round((t2 - t1) * 1e3, 2), flush=True)
for i in range(3):
    sync(); t0 = time.perf_counter(); poses = pipe.run()
    dist.all_gather(list(gathered.chunk(world)), poses.contiguous()); sync(); t2 = time.perf_counter()
    print(rank, i, 'run+gather (no sync between) ms', round((t2 - t0) * 1e3, 2), flush=True)
dist.barrier(); dist.destroy_process_group()
