"""CPU: the plain-C oracle (oracle/mtr_oracle.c) against the golden vectors of the real reference.
It is an independent restatement (no torch, scalar loops), so it agrees to rounding, not bitwise:
decode 1e-3 mm (reductions accumulate in double, elementwise ops in float),
reconstruct 4e-3 mm, pyramid 2 ulp, warp at the reference's own fp32 floor."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import c_oracle, cases, cpu_ref


@pytest.mark.parametrize('name', ['s256', 's256_legacy', 's256_peaked', 'l384', 'odd160', 's256_j1',
                                  's256_spike', 's256_d72'])
def test_c_decode_vs_golden(name):
    g = load_golden(f'heads_{name}')
    logits, J, cfg = cases.head_case(name)
    c2d, c3d = c_oracle.decode(logits.float().numpy(), J, cfg)
    assert np.abs(c3d - g['coords3d_rel']).max() <= 1e-3
    assert np.abs(c2d - g['coords2d']).max() <= 2e-4


@pytest.mark.parametrize('name', ['b64_j17', 'b5_j122_384', 'b1_j17', 'b8_legacy', 'b6_nomix', 'b4_outfov', 'b8_weak'])
def test_c_reconstruct_vs_golden(name):
    g = load_golden(f'recon_{name}')
    c2d, rel, K, cfg = cases.recon_case(name)
    out = c_oracle.reconstruct(c2d.numpy(), rel.numpy(), K.numpy(), cfg)
    assert np.abs(out - g['poses3d']).max() <= 4e-3
    assert float(np.linalg.norm(out - g['poses3d'], axis=-1).mean()) <= 1e-3


def test_c_pyramid_vs_oracle():
    img = cases.synth_images(2, 37, 53, 4)
    ref = cpu_ref.build_pyramid((img.float() / 255) ** 2.2)
    got = c_oracle.pyramid(img.numpy())
    for a, b in zip(got, ref):
        assert a.shape == tuple(b.shape) and np.abs(a - b.numpy()).max() <= 2.4e-7


@pytest.mark.parametrize('name', list(cases.WARP_CASES))
def test_c_warp_vs_golden(name):
    g = load_golden(f'warp_{name}')
    c = cases.warp_case(name)
    levels = cpu_ref.build_pyramid(c['images'])
    lv = cpu_ref.pyramid_level_index(c['crop_scales'])
    for i in range(6):
        kl = cpu_ref.corner_aligned_scale_mat(1 / 2 ** int(lv[i])) @ c['K'][i]
        out = c_oracle.warp(levels[lv[i]][c['image_ids'][i]].numpy(), kl.numpy(), c['hinv'][i].numpy(),
                            c['dist'][i].numpy(), c['res'])
        d = np.abs(out - g['crops'][i])
        assert d.max() <= 6e-5 and d.mean() <= 4e-6
