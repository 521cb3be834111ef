"""`python bench.py --gpus N` must start N ranks by itself (round-2 verdict: without a launcher around
it, it silently measured ONE GPU and printed n_gpus: 1 for a --gpus 8 command).

The gpurun box has one GPU and RCCL refuses two ranks on one device, so the two ranks share cuda:0
over gloo (MTR_BENCH_SHARED_DEVICE=1): this runs the N > 1 code path end to end -- self-spawn through
torch.distributed.run, rank binding, sharding of internal batches (multiperson_model.py:189-220's unit),
barrier + max-over-ranks timing, the single gather -- but is not a scaling measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT_KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'}


def run_bench(*flags, shared=True, timeout=1200):
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    if shared:
        env['MTR_BENCH_SHARED_DEVICE'] = '1'
    else:
        env.pop('MTR_BENCH_SHARED_DEVICE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *flags], capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    return r, (json.loads(lines[-1]) if lines else None), lines


def keep(name, line):
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name), 'w') as f:
            f.write(json.dumps(line) + '\n')
    except OSError:
        pass


def test_gpus_2_on_one_gpu_refuses_instead_of_measuring_one(hip_lib):
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('a multi-GPU box: nothing to refuse')
    r, line, _ = run_bench('--gpus', '2', '--steps', '1', '--warmup', '0', '--quick', shared=False, timeout=300)
    assert r.returncode != 0 and line is None
    assert 'refusing' in (r.stderr + r.stdout)


def test_launcher_and_flag_disagreeing_is_an_error(hip_lib):
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--quick', '--steps', '1'],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and 'disagree' in (r.stderr + r.stdout)


def test_gpus_2_weak_scaling_line(hip_lib):
    r1, one, _ = run_bench('--gpus', '1', '--steps', '3', '--warmup', '1', '--quick', shared=False)
    assert r1.returncode == 0 and one is not None, r1.stderr[-2000:]
    r, two, lines = run_bench('--gpus', '2', '--steps', '3', '--warmup', '1', '--quick')
    assert r.returncode == 0 and two is not None, r.stderr[-2000:]
    assert len(lines) == 1, 'only rank 0 prints the line'
    assert CONTRACT_KEYS <= set(two)
    assert two['n_gpus'] == 2 and one['n_gpus'] == 1
    assert two['scaling'] == 'weak' and two['config']['global_batch'] == 2 * one['config']['global_batch']
    m = two['multi_gpu']
    assert len(m['per_rank_ms_per_step']) == 2 and m['backend'] == 'gloo'
    assert abs(max(m['per_rank_ms_per_step']) - two['ms_per_step']) < 1e-2  # the max over ranks is the job's time
    # two ranks SHARE one GPU here: the job moves twice the crops in about twice the time, i.e. about the
    # one-GPU rate (graphs of the two processes interleave; the host-side gloo gather adds a little)
    assert 0.45 * one['value'] <= two['value'] <= 1.35 * one['value'], (one['value'], two['value'])
    keep('bench_gpus2_shared_device_config1.json', two)


def test_gpus_2_strong_scaling_line(hip_lib):
    r, two, lines = run_bench('--gpus', '2', '--steps', '2', '--warmup', '1', '--quick', '--config', '2',
                              '--total-crops', '128')
    assert r.returncode == 0 and two is not None, r.stderr[-2000:]
    assert len(lines) == 1 and CONTRACT_KEYS <= set(two)
    assert two['n_gpus'] == 2 and two['scaling'] == 'strong'
    assert two['config']['global_batch'] == 128 and 'configs[2]' in two['config']['workload']
    assert len(two['multi_gpu']['per_rank_ms_per_step']) == 2
    assert two['value'] > 0
    keep('bench_gpus2_shared_device_config2.json', two)
