"""GPU parity of K9, the detector pre-processing in front of the hot path (row f.3), through the
C-ABI: network input tensor and rescaled boxes vs the golden vectors recorded from the reference's
own PersonDetector.forward (tests/golden/detpre_*.npz) and vs the oracle at sizes the goldens do not
hold.  The linear-light resize mirrors aten's CPU kernels operation for operation; what is left is
the LUT's (v/255)**2.2 (<= 1 ulp from torch's pow) and the final **(1/2.2): bound 6e-7 absolute on
values in [0, 1]."""
import pytest
import torch

from conftest import load_golden
from oracle import cases, cpu_ref

pytestmark = pytest.mark.gpu
TOL = 6e-7


@pytest.mark.parametrize('name', list(cases.DETPRE_CASES))
def test_detector_preprocess_vs_golden(name, hip_lib):
    from metrabs_amd.multiperson.person_detector import PersonDetector
    g = load_golden(f'detpre_{name}')
    c = cases.detpre_case(name)
    fed = {}

    def network(x, threshold, nms_iou_threshold, max_detections):
        fed['x'] = x
        assert (threshold, nms_iou_threshold, max_detections) == (0.3, 0.7, 150)
        return [b.cuda() for b in c['net_boxes']]

    boxes = PersonDetector(network)(c['images'].cuda(), 0.3, 0.7, 150)
    x = fed['x'].cpu()
    assert list(x.shape) == list(g['network_input_shape'])
    d = (x[:, :, ::7, ::5] - torch.from_numpy(g['network_input_sample'])).abs()
    print(f'[parity] detector input {name}: max-abs {float(d.max()):.2e} mean {float(d.mean()):.2e}')
    assert float(d.max()) <= TOL
    assert [len(b) for b in boxes] == list(g['n_box'])
    got = torch.cat(boxes).cpu()
    assert torch.equal(got, torch.from_numpy(g['boxes'])), float((got - torch.from_numpy(g['boxes'])).abs().max())


@pytest.mark.parametrize('shape', [(1, 1080, 1920), (2, 720, 1280), (1, 2160, 3840), (1, 333, 1999),
                                   (1, 64, 416), (2, 8, 8), (1, 417, 200)])
def test_detector_preprocess_vs_oracle_full_tensor(shape, hip_lib):
    """Full-size frames (1080p = BASELINE configs, 4K), extreme aspect ratios, tiny frames: every
    element against the oracle, and the pad region exactly 0.5."""
    from metrabs_amd import kernels
    n, h, w = shape
    img = cases.synth_images(n, h, w, 31)
    with torch.inference_mode():
        ref, m = cpu_ref.detector_preprocess(img)
    x, g = kernels.detector_preprocess(img.cuda())
    x = x.cpu()
    assert x.shape == ref.shape
    d = (x - ref).abs()
    print(f'[parity] detector input {shape}: max-abs {float(d.max()):.2e}')
    assert float(d.max()) <= TOL
    inner = torch.zeros(g.out_h, g.out_w, dtype=torch.bool)
    inner[g.pad_top:g.pad_top + g.target_h, g.pad_left:g.pad_left + g.target_w] = True
    assert bool((x[:, :, ~inner] == 0.5).all())


def test_binary_frame_isolates_the_final_pow(hip_lib):
    """A frame of only 0 and 255 has LUT values exactly 0.0 / 1.0, so the kernel's linear-light
    resize equals aten's bit for bit (same weights, same fma order) and the only difference left is
    the last operation, pow(x, 1/2.2): torch's vectorised CPU pow and the device powf round
    differently on ~10 % of the values, never by more than 1 ulp."""
    from metrabs_amd import kernels
    g = cases.gen(77)
    img = (torch.randint(0, 2, (1, 3, 540, 960), generator=g) * 255).to(torch.uint8)
    with torch.inference_mode():
        ref, _ = cpu_ref.detector_preprocess(img)
    x, _ = kernels.detector_preprocess(img.cuda())
    x = x.cpu()
    same = float((x == ref).float().mean())
    print(f'[parity] binary frame: {same * 100:.3f} % bit-identical, max {float((x - ref).abs().max()):.2e}')
    assert same >= 0.8 and float((x - ref).abs().max()) <= 6e-8


def test_shrink_factor_limit_is_reported(hip_lib):
    from metrabs_amd import kernels
    img = torch.zeros(1, 3, 16, 9000, dtype=torch.uint8).cuda()  # 21.6x shrink
    with pytest.raises(RuntimeError):
        kernels.detector_preprocess(img)
