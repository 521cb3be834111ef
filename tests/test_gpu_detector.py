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
                                   (1, 64, 416), (2, 8, 8), (1, 417, 200), (1, 200, 6000), (1, 4500, 300)])
@pytest.mark.parametrize('kernel', ['tile', 'stream'])
def test_detector_preprocess_vs_oracle_full_tensor(shape, kernel, hip_lib):
    """Full-size frames (1080p = BASELINE configs, 4K), extreme aspect ratios (14x shrink: the
    40-tap instantiation), tiny frames, tensor sizes that are not multiples of 16 (byte-wise tail):
    every element against the oracle, and the pad region exactly 0.5."""
    from metrabs_amd import kernels
    n, h, w = shape
    img = cases.synth_images(n, h, w, 31)
    with torch.inference_mode():
        ref, m = cpu_ref.detector_preprocess(img)
    if kernel == 'stream' and img.numel() % 16:
        # the streaming kernel loads whole 16-byte vectors: such tensors are the tile kernel's
        with pytest.raises(RuntimeError):
            kernels.detector_preprocess(img.cuda(), kernel='stream')
        x, g = kernels.detector_preprocess(img.cuda())  # ('auto' falls back)
        assert torch.equal(x, kernels.detector_preprocess(img.cuda(), kernel='tile')[0])
        return
    x, g = kernels.detector_preprocess(img.cuda(), kernel=kernel)
    x = x.cpu()
    assert x.shape == ref.shape
    d = (x - ref).abs()
    print(f'[parity] detector input {shape} ({kernel}): max-abs {float(d.max()):.2e}')
    assert float(d.max()) <= TOL
    inner = torch.zeros(g.out_h, g.out_w, dtype=torch.bool)
    inner[g.pad_top:g.pad_top + g.target_h, g.pad_left:g.pad_left + g.target_w] = True
    assert bool((x[:, :, ~inner] == 0.5).all())


@pytest.mark.parametrize('shape', [(8, 1080, 1920), (3, 720, 1280), (1, 2160, 3840), (2, 480, 640), (1, 64, 416),
                                   (5, 8, 8), (1, 200, 6000), (1, 4500, 300), (2, 1088, 1920), (1, 256, 416),
                                   (4, 250, 400), (1, 1080, 1928), (7, 360, 648)])
def test_stream_and_tile_kernels_give_identical_bits(shape, hip_lib):
    """The streaming kernel (strip walked down the frame, loading wave, ring of filtered rows) and the
    tile kernel evaluate the same weights and the same fma chains from the same gamma table: every
    output bit is equal, for shrinking (12 / 24 / 40 tap instantiations), enlarging and 1:1 frames,
    unit boundaries inside a plane (8 x 1080p: three units per plane; one frame: more) and rows
    whose staged start is not 16-byte aligned (W = 1928)."""
    from metrabs_amd import kernels
    n, h, w = shape
    img = cases.synth_images(n, h, w, 5).cuda()
    tile, g = kernels.detector_preprocess(img, kernel='tile')
    stream, _ = kernels.detector_preprocess(img, kernel='stream')
    auto, _ = kernels.detector_preprocess(img)  # (whichever of the two the launch size picks)
    assert torch.equal(tile, stream), float((tile - stream).abs().max())
    assert torch.equal(auto, stream)
    # a second call into a poisoned buffer: every element (pad included) is written
    out = torch.full_like(stream, float('nan'))
    kernels.detector_preprocess(img, geom=g, out=out, kernel='stream')
    assert torch.equal(out, stream)


def test_stream_kernel_random_shapes(hip_lib):
    """60 seeded random frame shapes (8 .. 2600 px a side, 1 - 3 frames; enlarging, shrinking up to the
    40-tap limit, extreme aspect ratios): wherever the streaming kernel is available its output is
    bit-identical to the tile kernel's; every fifth shape is also checked against the oracle."""
    import random
    from metrabs_amd import kernels
    rng = random.Random(20260925)
    n_stream = 0
    for it in range(60):
        n = rng.choice([1, 1, 2, 3])
        h = rng.choice([rng.randint(8, 200), rng.randint(200, 1200), rng.randint(1200, 2600)])
        w = rng.choice([rng.randint(8, 200), rng.randint(200, 1200), rng.randint(1200, 2600)])
        if rng.random() < 0.7:
            w = max(16, w // 16 * 16)  # (a tensor of a multiple of 16 bytes: the streaming kernel's domain)
        img = cases.synth_images(n, h, w, 100 + it).cuda()
        try:
            tile, g = kernels.detector_preprocess(img, kernel='tile')
        except RuntimeError:
            continue  # (beyond the 19x shrink limit)
        if img.numel() % 16 == 0:
            try:
                stream, _ = kernels.detector_preprocess(img, kernel='stream')
            except RuntimeError:
                stream = None  # (its LDS does not fit this shape: `auto` takes the tile kernel)
                assert torch.equal(kernels.detector_preprocess(img)[0], tile)
            if stream is not None:
                assert torch.equal(tile, stream), (n, h, w, float((tile - stream).abs().max()))
                n_stream += 1
        if it % 5 == 0:
            with torch.inference_mode():
                ref, _ = cpu_ref.detector_preprocess(img.cpu())
            assert float((tile.cpu() - ref).abs().max()) <= TOL, (n, h, w)
    assert n_stream >= 30


def test_binary_frame_isolates_the_final_pow(hip_lib):
    """A frame of only 0 and 255 has LUT values exactly 0.0 / 1.0, so the kernel's linear-light
    resize equals aten's bit for bit (same weights, same fma order) and the only difference left is
    the last operation, pow(x, 1/2.2): the kernel's is correctly rounded (evaluated in double), torch's
    vectorised CPU pow is for ~98 % of the values and never off by more than 1 ulp (with the device
    library's powf, the first version, the two agreed on ~90 %)."""
    from metrabs_amd import kernels
    g = cases.gen(77)
    img = (torch.randint(0, 2, (1, 3, 540, 960), generator=g) * 255).to(torch.uint8)
    with torch.inference_mode():
        ref, _ = cpu_ref.detector_preprocess(img)
    x, _ = kernels.detector_preprocess(img.cuda())
    x = x.cpu()
    same = float((x == ref).float().mean())
    print(f'[parity] binary frame: {same * 100:.3f} % bit-identical, max {float((x - ref).abs().max()):.2e}')
    assert same >= 0.95 and float((x - ref).abs().max()) <= 6e-8


def test_shrink_factor_limit_is_reported(hip_lib):
    from metrabs_amd import kernels
    img = torch.zeros(1, 3, 16, 9000, dtype=torch.uint8).cuda()  # 21.6x shrink
    with pytest.raises(RuntimeError):
        kernels.detector_preprocess(img)


def test_detect_poses_runs_detector_then_the_hot_path(hip_lib):
    """detect_poses_batched (multiperson_model.py:56-75 of the reference) = PersonDetector + the
    per-crop path: with a stub network the result must equal estimate on the rescaled boxes."""
    from test_gpu_e2e import build_estimator
    from metrabs_amd.multiperson.person_detector import PersonDetector
    case = cases.e2e_case('aug5')
    est = build_estimator(case, fused_head=True)
    n_img, _, h, w = case['images'].shape
    geom = cpu_ref.detector_target_size(h, w)
    g = cases.gen(99)
    net_boxes = []
    for i in range(n_img):  # boxes well inside the network frame, one image without detections
        k = 0 if i == 1 else 2
        x1 = geom['pad_left'] + torch.rand(k, generator=g) * geom['target_w'] * 0.4
        y1 = geom['pad_top'] + torch.rand(k, generator=g) * geom['target_h'] * 0.3
        net_boxes.append(torch.stack([x1, y1, x1 + 0.3 * geom['target_w'], y1 + 0.6 * geom['target_h'],
                                      0.5 + 0.5 * torch.rand(k, generator=g)], dim=1).float())
    est.detector = PersonDetector(lambda x, t, i, m: [b.cuda() for b in net_boxes])
    with torch.inference_mode():
        got = est.detect_poses_batched(case['images'], case['K'], case['dist'], case['extr'], case['world_up'],
                                       55, case['ibs'], case['aa'], case['num_aug'], case['average_aug'], '',
                                       0.3, 0.7, 150, False, False)
        boxes = [cpu_ref.detector_scale_boxes(b, geom) for b in net_boxes]
        want = est._estimate_poses_batched(
            case['images'], boxes, case['K'], case['dist'], case['extr'], case['world_up'], 55, case['ibs'],
            case['aa'], case['num_aug'], case['average_aug'], '', False)
    assert [len(b) for b in got['boxes']] == [len(b) for b in net_boxes]
    for a, b in zip(got['boxes'], boxes):
        assert torch.equal(a.cpu(), b)
    for a, b in zip(got['poses3d'], want['poses3d']):
        assert torch.equal(a, b)
