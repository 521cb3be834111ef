"""GPU: Pose3dEstimator sharded over TWO ranks that share cuda:0 (gloo rendezvous; RCCL refuses two
ranks on one device and the gpurun box has one GPU) against the single-rank call in the same
processes -- the code path the 8-GPU run takes, end to end: per-rank pyramids of the frames a
rank's boxes reference, the crop pipeline, K7, one all-gather.

* shard_across_ranks=True: whole internal batches dealt round-robin (multiperson_model.py:189-220 is
  the unit): every internal batch is computed by exactly the kernels of the single-rank run, so the
  result must be BIT-EQUAL.
* shard_across_ranks='exact_monolithic': every internal batch cut into one slice per rank, the
  three reconstruction moments all-reduced: equal to the single-rank result up to the order of the
  f64 moment sums (and whatever MIOpen does differently at another batch size): <= 1e-3 mm."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


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
    out = {}
    for name in ('aug5', 'aug4_dist12', 'aug5_dist_aa2'):
        if name not in cases.E2E_CASES:
            continue
        case = cases.e2e_case(name)
        est = build_estimator(case, 'auto')   # the default head dispatch: a static rule, the same on every rank and slice
        args = (case['images'], case['boxes'], case['K'], case['dist'], case['extr'], case['world_up'],
                55, case['ibs'], case['aa'], case['num_aug'], case['average_aug'], '', False)
        with torch.inference_mode():
            single = est._estimate_poses_batched(*args)
            est.shard_across_ranks = True
            sharded = est._estimate_poses_batched(*args)
            est.shard_across_ranks = 'exact_monolithic'
            exact = est._estimate_poses_batched(*args)
            # one big internal batch cut over the ranks: the moments of the WHOLE batch
            big = list(args)
            big[7] = 4096
            est.shard_across_ranks = False
            single_big = est._estimate_poses_batched(*big)
            est.shard_across_ranks = 'exact_monolithic'
            exact_big = est._estimate_poses_batched(*big)
        cat = lambda r, k: torch.cat(r[k]).cpu().numpy()
        out[name] = {k: (cat(single, k), cat(sharded, k), cat(exact, k), cat(single_big, k), cat(exact_big, k))
                     for k in ('poses3d', 'poses2d')}
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_single_rank_result(hip_lib):
    world = 2
    ctx = mp.get_context('spawn')
    os.environ['PYTHONPATH'] = os.pathsep.join(
        [os.path.dirname(os.path.abspath(__file__)), os.environ.get('PYTHONPATH', '')])
    for attempt in range(3):
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
    assert results[0][1], 'no e2e case ran'
    for rank, out in results:
        for name, res in out.items():
            for key, (single, sharded, exact, single_big, exact_big) in res.items():
                assert single.shape == sharded.shape == exact.shape and len(single) > 0
                assert np.array_equal(single, sharded), (rank, name, key, np.abs(single - sharded).max())
                for a, b in ((single, exact), (single_big, exact_big)):
                    d = np.abs(a - b)
                    if key == 'poses3d':
                        assert d.max() <= 1e-3, (rank, name, key, d.max())   # mm
                    else:
                        # px: a random-weight head puts some joints at near-zero depth, where x / z turns the
                        # 1e-4 mm between a slice and the whole batch into hundredths of a pixel (seen once in
                        # eight runs: 0.03 px on one joint) -- the bulk is gated, the worst joint bounded
                        assert np.quantile(d, 0.95) <= 1e-4 and d.max() <= 0.5, (rank, name, key, d.max())
    # both ranks hold the same gathered result
    for name in results[0][1]:
        for key in ('poses3d', 'poses2d'):
            for a, b in zip(results[0][1][name][key], results[1][1][name][key]):
                assert np.array_equal(a, b)
