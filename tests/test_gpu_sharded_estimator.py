"""GPU: Pose3dEstimator sharded over TWO ranks that share cuda:0 (gloo rendezvous; RCCL refuses two
ranks on one device and the gpurun box has one GPU) against the single-rank call in the same
processes -- the code path the 8-GPU run takes, end to end: per-rank pyramids of the frames a
rank's boxes reference, the crop pipeline, K7, one all-gather.

* shard_across_ranks=True ('round_robin'): whole internal batches dealt round-robin
  (multiperson_model.py:189-220 is the unit): every internal batch is computed by exactly the kernels of
  the single-rank run, so the result must be BIT-EQUAL -- asserted on two crop models: the tiny conv
  backbone + random head of the e2e goldens (MIOpen in the loop) and the plausible-pose model below.
* shard_across_ranks='exact_monolithic' ('exact': the case's own internal batch size; 'exact_big': ONE
  internal batch of 4096 cut over the ranks): every internal batch cut into one slice per rank, the
  three reconstruction moments all-reduced: equal to the single-rank result up to the order of the f64
  moment sums, i.e. an f32 ulp of the reference depth here and there: <= 1e-3 mm, <= 1e-4 px.

Round 6: the 'exact' comparisons run on cases.PlausiblePoseBackbone -- people 1 - 4.5 m
from the camera -- instead of the random head whose joints sit at near-zero depth, where x / z turned
1e-4 mm into hundredths of a pixel once in eight runs; one worker pair computes everything once
(module fixture) and every (mode, case) is its own test, so one failure cannot hide the others."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CASES = ('aug5', 'aug4_dist12', 'aug5_dist_aa2')
KEYS = ('poses3d', 'poses2d')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK='0')
    import torch.distributed as dist
    from metrabs_amd import distributed
    from oracle import cases
    from test_gpu_e2e import build_estimator
    torch.cuda.set_device(0)
    distributed.init_from_env(backend='gloo')
    cat = lambda r, k: torch.cat(r[k]).cpu().numpy()
    out = {}
    for name in CASES:
        base = cases.e2e_case(name)
        for model in ('random', 'plausible'):
            case = cases.with_plausible_pose_model(base) if model == 'plausible' else base
            est = build_estimator(case, 'auto')   # the default head dispatch: a static rule, the same on every rank and slice
            args = (case['images'], case['boxes'], case['K'], case['dist'], case['extr'], case['world_up'],
                    55, case['ibs'], case['aa'], case['num_aug'], case['average_aug'], '', False)
            runs = {}
            with torch.inference_mode():
                est.shard_across_ranks = False
                runs['single'] = est._estimate_poses_batched(*args)
                est.shard_across_ranks = True
                runs['round_robin'] = est._estimate_poses_batched(*args)
                if model == 'plausible':
                    est.shard_across_ranks = 'exact_monolithic'
                    runs['exact'] = est._estimate_poses_batched(*args)
                    # one big internal batch cut over the ranks: the moments of the WHOLE batch
                    big = list(args)
                    big[7] = 4096
                    est.shard_across_ranks = False
                    runs['single_big'] = est._estimate_poses_batched(*big)
                    est.shard_across_ranks = 'exact_monolithic'
                    runs['exact_big'] = est._estimate_poses_batched(*big)
            out[name, model] = {run: {k: cat(r, k) for k in KEYS} for run, r in runs.items()}
            if model == 'plausible':
                E = case['extr'][0].numpy().astype(np.float64)
                cam = out[name, model]['single']['poses3d'].reshape(-1, 3) @ E[:3, :3].T + E[:3, 3]
                out[name, model]['camera_z'] = cam[:, 2]
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope='module')
def two_rank_results(hip_lib):
    world = 2
    ctx = mp.get_context('spawn')
    os.environ['PYTHONPATH'] = os.pathsep.join(
        [os.path.dirname(os.path.abspath(__file__)), os.environ.get('PYTHONPATH', '')])
    for attempt in range(3):   # (a rendezvous that does not come up, not a numeric retry: results are asserted once)
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            results = [q.get(timeout=300) for _ in range(world)]
        except Exception:
            results = None
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        if results is not None and all(p.exitcode == 0 for p in procs):
            break
    else:
        pytest.fail('world-size-2 run on cuda:0 failed three times')
    return dict(results)


@pytest.mark.parametrize('model', ['random', 'plausible'])
@pytest.mark.parametrize('name', CASES)
def test_round_robin_sharding_is_bit_equal_to_the_single_rank_result(two_rank_results, name, model):
    for rank, out in two_rank_results.items():
        res = out[name, model]
        for key in KEYS:
            single, sharded = res['single'][key], res['round_robin'][key]
            assert single.shape == sharded.shape and len(single) > 0
            assert np.array_equal(single, sharded), (rank, name, key, np.abs(single - sharded).max())


def test_the_plausible_pose_model_keeps_joints_away_from_the_camera_plane(two_rank_results):
    """What makes the px gate below meaningful: no joint of the comparison cases is nearer than 0.5 m."""
    for name in CASES:
        z = two_rank_results[0][name, 'plausible']['camera_z']
        assert z.min() > 500 and z.max() < 8000, (name, z.min(), z.max())


@pytest.mark.parametrize('mode', ['exact', 'exact_big'])
@pytest.mark.parametrize('name', CASES)
def test_exact_monolithic_slices_match_the_single_rank_batch(two_rank_results, name, mode):
    for rank, out in two_rank_results.items():
        res = out[name, 'plausible']
        for key in KEYS:
            a, b = res['single' if mode == 'exact' else 'single_big'][key], res[mode][key]
            assert a.shape == b.shape and len(a) > 0
            d = np.abs(a - b)
            print(f'[sharded] rank {rank} {name} {mode} {key}: max {d.max():.2e}, differing {np.count_nonzero(d)} of {d.size}')
            if key == 'poses3d':
                assert d.max() <= 1e-3, (rank, name, mode, key, d.max())   # mm
            else:
                assert np.quantile(d, 0.95) <= 1e-4 and d.max() <= 1e-3, (rank, name, mode, key, d.max())   # px


@pytest.mark.parametrize('name', CASES)
def test_both_ranks_hold_the_same_gathered_result(two_rank_results, name):
    for model in ('random', 'plausible'):
        r0, r1 = two_rank_results[0][name, model], two_rank_results[1][name, model]
        for run in r0:
            if run == 'camera_z':
                continue
            for key in KEYS:
                assert np.array_equal(r0[run][key], r1[run][key]), (name, model, run, key)
