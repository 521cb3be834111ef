"""CPU: no kernel of libmetrabs_hip.so uses scratch memory.  The code object metadata of every HIP
source (hipcc -S --cuda-device-only, the build's own flags; cross-compiles without a GPU) must
report private_segment_fixed_size 0 and no spilled VGPRs -- round 1's f32 head for 12x12 and
16x16 maps spilled up to 339 VGPRs (1.1 KB of scratch per lane) and ran at 10 % of the MFMA peak."""
from concurrent.futures import ThreadPoolExecutor

import pytest


@pytest.fixture(scope='module')
def resources():
    from metrabs_amd import build
    srcs = build.sources()
    with ThreadPoolExecutor(max_workers=8) as ex:
        return dict(zip(srcs, ex.map(build.kernel_resources, srcs)))


def test_no_kernel_spills_or_uses_scratch(resources):
    seen = 0
    for src, kernels in resources.items():
        for k in kernels:
            seen += 1
            assert k['private_segment_fixed_size'] == 0, (src, k)
            assert k['vgpr_spill_count'] == 0, (src, k)  # (SGPRs parked in VGPR lanes are not scratch)
    assert seen >= 200  # every template instantiation of every source was looked at


def test_head_kernels_fit_their_occupancy_targets(resources):
    """head_rt_kernel<3, *> (small launches: one workgroup per CU) may use the whole register file;
    head_rt_kernel<5, *> is built for two workgroups per CU (__launch_bounds__(256, 2)): <= 256
    VGPRs.  The 16-bit joint-group kernels stay at or below 256 as well."""
    heads = [k for src in ('head_rt.hip', 'head_fused.hip') for k in resources[src]
             if 'head_rt_kernel' in k['name'] or 'head_fused16' in k['name']]
    assert len(heads) >= 50
    for k in heads:
        assert k['vgpr_count'] + k['agpr_count'] <= 512
        if 'head_rt_kernelILi5' in k['name']:
            assert k['vgpr_count'] <= 256, k
