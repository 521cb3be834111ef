import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)


def same_cpu_as_golden(g):
    """Golden vectors are bit-reproducible only with the torch build + CPU kernels (MKL/oneDNN
    dispatch) they were minted with."""
    import torch
    try:
        with open('/proc/cpuinfo') as f:
            cpu = [l.split(':', 1)[1].strip() for l in f if l.startswith('model name')][0]
    except (OSError, IndexError):
        cpu = ''
    return (str(g['torch_version']) == torch.__version__ and str(g['cpu']) == cpu
            and int(g['num_threads']) == torch.get_num_threads())


@pytest.fixture(scope='session')
def hip_lib():
    """The C-ABI library; parity tests call THROUGH it.  Fails loudly if it is missing."""
    from metrabs_amd import _lib
    return _lib.load()
