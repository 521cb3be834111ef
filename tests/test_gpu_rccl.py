"""RCCL on the box there is: a process group of ONE rank with backend "nccl" (= RCCL on ROCm) on cuda:0.
It loads librccl, builds a communicator and runs the path's two collectives on DEVICE tensors with no
host hop -- the all-gather of the poses (metrabs_amd.distributed.gather_ranges) eagerly and captured in
a HIP graph (alone, and as the after_step hook inside the captured crop pipeline), and the 3-double moment
all-reduce -- so that the driver's 8-GPU run is not RCCL's first execution of this code.  Run in a child
process: a process group is process-global state the other tests must not inherit."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from metrabs_amd import distributed
torch.cuda.set_device(0)
rank, world, _ = distributed.init_from_env(backend='nccl', force_group=True)
assert dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == 1
out = {'backend': dist.get_backend(), 'world': dist.get_world_size()}
dev = torch.device('cuda', 0)

# 1. the pose gather on device tensors, eager
local = torch.randn(37, 17, 5, device=dev)
ranges = [[(0, 20), (20, 37)]]
got = distributed.gather_ranges(local, ranges, 37, always=True)
assert got.is_cuda and got.data_ptr() != local.data_ptr() and torch.equal(got, local)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    distributed.gather_ranges(local, ranges, 37, always=True)
torch.cuda.synchronize()
out['gather_ranges_us'] = (time.perf_counter() - t0) / 50 * 1e6

# 2. the moment all-reduce on a device f64 triple
m = torch.tensor([1.5, 2.5, 64.0], dtype=torch.float64, device=dev)
distributed.allreduce_moments(m, always=True)
assert m.tolist() == [1.5, 2.5, 64.0]

# 3. all_gather_into_tensor captured in a HIP graph
src = torch.randn(64, 17, 3, device=dev)
dst = torch.zeros_like(src)
dist.all_gather_into_tensor(dst, src)   # warm-up outside the capture (communicator, buffers)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode='thread_local'):  # (the watchdog thread polls events meanwhile)
        dist.all_gather_into_tensor(dst, src)
    src.normal_()
    dst.zero_()
    g.replay()
    torch.cuda.synchronize()
    out['graph_capture'] = bool(torch.equal(dst, src))
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    out['graph_gather_us'] = (time.perf_counter() - t0) / 200 * 1e6
except Exception as e:
    out['graph_capture'] = False
    out['graph_capture_error'] = str(e)[:300]
t0 = time.perf_counter()
for _ in range(200):
    dist.all_gather_into_tensor(dst, src)
torch.cuda.synchronize()
out['eager_gather_us'] = (time.perf_counter() - t0) / 200 * 1e6

# 4. the sharded estimator end to end with the collective forced, and the hook inside the pipeline's graph
from oracle import cases
from test_gpu_e2e import build_estimator
case = cases.e2e_case('aug5')
est = build_estimator(case, 'auto')
args = (case['images'], case['boxes'], case['K'], case['dist'], case['extr'], case['world_up'], 55, case['ibs'],
        case['aa'], case['num_aug'], case['average_aug'], '', False)
single = est._estimate_poses_batched(*args)
est.shard_across_ranks, est.force_collective = True, True
sharded = est._estimate_poses_batched(*args)
assert torch.equal(torch.cat(single['poses3d']), torch.cat(sharded['poses3d']))
out['estimator_with_forced_gather'] = True
if out['graph_capture']:
    from metrabs_amd.pipeline import GraphedCropPipeline
    est.shard_across_ranks = False
    pipe = GraphedCropPipeline(est, len(case['images']), case['images'].shape[2], case['images'].shape[3], 3, num_aug=2)
    pipe.images.copy_(case['images'])
    pipe.boxes.copy_(torch.cat(case['boxes'])[:3, :4])
    gathered = torch.zeros(3, 17, 3, device=dev)
    pipe.after_step = lambda poses: dist.all_gather_into_tensor(gathered, poses.contiguous())
    pipe.capture()
    gathered.zero_()
    res = pipe.run()
    torch.cuda.synchronize()
    out['pipeline_graph_with_gather'] = bool(torch.equal(gathered, res) and float(res.abs().max()) > 0)
print('RCCL_RESULT ' + json.dumps(out))
dist.barrier()
dist.destroy_process_group()
'''


def test_one_rank_rccl_group_runs_the_paths_collectives(hip_lib):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1',
               LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0',
               PYTHONPATH=os.pathsep.join([ROOT, os.environ.get('PYTHONPATH', '')]))
    r = subprocess.run([sys.executable, '-c', f'ROOT = {ROOT!r}\n' + CHILD], env=env, capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith('RCCL_RESULT ')]
    assert line, r.stdout[-2000:]
    out = json.loads(line[-1][len('RCCL_RESULT '):])
    print('[rccl]', out)
    assert out['backend'] == 'nccl' and out['world'] == 1 and out['estimator_with_forced_gather']
    # graph capture of the collective is reported, not required (the eager gather is the default path)
    if out['graph_capture']:
        assert out.get('pipeline_graph_with_gather', False)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'rccl_one_rank.json'), 'w') as f:
        json.dump(out, f)
