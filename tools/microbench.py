#!/usr/bin/env python
"""Per-kernel micro-benchmarks of libmetrabs_hip.so against their rooflines (HIP events on the launch
stream, random non-zero data, >= 50 iterations after warm-up).

    python tools/microbench.py [decode] [head] [warp] [pyramid] [recon]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402

HBM = 8.0e12


def timeit(fn, iters=50, warm=10, graph=True):
    """Average duration of one launch: `iters` launches captured in ONE HIP graph (no Python /
    ctypes / allocator time between them), replayed 3 times between two HIP events.  graph=False
    (wrappers that copy host lists to the device): an eager loop between two events."""
    if not graph:
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e-3 / iters
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st), torch.inference_mode():
        for _ in range(min(warm, 3)):
            fn()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st, capture_error_mode='thread_local'):
            for _ in range(iters):
                fn()
        g.replay()
        st.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(3):
            g.replay()
        b.record(st)
        st.synchronize()
    torch.cuda.current_stream().wait_stream(st)
    return a.elapsed_time(b) * 1e-3 / (3 * iters)


def bench_decode():
    out = []
    g = torch.Generator(device='cuda').manual_seed(0)
    for name, B, J, D, H, W, dt in [('s256 fp32 (roofline shape)', 32768, 17, 8, 8, 8, torch.float32),
                                    ('l384 J=122 fp32', 2048, 122, 8, 12, 12, torch.float32),
                                    ('D=72 stress', 4096, 17, 72, 8, 8, torch.float32),
                                    ('s256 fp16', 65536, 17, 8, 8, 8, torch.float16),
                                    ('s256 B=64', 64, 17, 8, 8, 8, torch.float32)]:
        cfg = MetrabsConfig(depth=D, proc_side=H * 32)
        x = torch.randn(B, J * (1 + D), H, W, device='cuda', generator=g).to(dt)
        o = (torch.empty(B, J, 2, device='cuda'), torch.empty(B, J, 3, device='cuda'))
        t = timeit(lambda: kernels.softargmax_decode(x, J, cfg, out=o))
        nbytes = x.numel() * x.element_size() + B * J * 20
        out.append(dict(kernel='decode', case=name, us=round(t * 1e6, 1), GBps=round(nbytes / t / 1e9, 1),
                        frac_hbm=round(nbytes / t / HBM, 3)))
        del x
    return out


def bench_head():
    out = []
    g = torch.Generator(device='cuda').manual_seed(0)
    core = 'row-tile (f32) / joint-group MFMA (16-bit)'
    for name, B, C, J, H, dt, nhwc in [('cfg2 B=64 f32', 64, 1280, 17, 8, torch.float32, False),
                                       ('cfg2 B=64 f32 nhwc', 64, 1280, 17, 8, torch.float32, True),
                                       ('cfg2 B=64 f16', 64, 1280, 17, 8, torch.float16, False),
                                       ('B=1024 f32', 1024, 1280, 17, 8, torch.float32, False),
                                       ('B=1024 f32 nhwc', 1024, 1280, 17, 8, torch.float32, True),
                                       ('B=1024 f16 feats', 1024, 1280, 17, 8, torch.float16, False),
                                       ('B=1024 f16 nhwc', 1024, 1280, 17, 8, torch.float16, True),
                                       ('HW=100 B=64 f32', 64, 1280, 17, 10, torch.float32, False),
                                       ('HW=100 B=512 f32', 512, 1280, 17, 10, torch.float32, False),
                                       ('HW=100 B=512 f16', 512, 1280, 17, 10, torch.float16, False),
                                       ('cfg3 B=32 384px f32', 32, 1280, 17, 12, torch.float32, False),
                                       ('cfg3 B=256 384px f32', 256, 1280, 17, 12, torch.float32, False),
                                       ('cfg5 B=32 J=122 f16', 32, 1280, 122, 12, torch.float16, False),
                                       ('cfg5 B=256 J=122 f16', 256, 1280, 122, 12, torch.float16, False)]:
        cfg = MetrabsConfig(proc_side=H * 32)
        feat = torch.randn(B, C, H, H, device='cuda', generator=g).to(dt)
        if nhwc:
            feat = feat.contiguous(memory_format=torch.channels_last)
        w = torch.randn(J * 9, C, device='cuda', generator=g) * 0.03
        b = torch.zeros(J * 9, device='cuda')
        packed = kernels.head_pack_weights(w, b, J, 8, dt)
        o = (torch.empty(B, J, 2, device='cuda'), torch.empty(B, J, 3, device='cuda'))
        t = timeit(lambda: kernels.head_fused(feat, packed, C, J, cfg, out=o))
        flops = 2.0 * C * J * 9 * H * H * B
        # f32 features: f32-input MFMA (157.3 TFLOP/s dense); 16-bit: f16 / bf16 MFMA (2.5 PFLOP/s
        # dense, MI355X_MICROARCH.md) -- there the kernel is staging / HBM bound, see frac_hbm
        peak = 157.3e12 if dt == torch.float32 else 2.5e15
        nbytes = feat.numel() * feat.element_size()
        out.append(dict(kernel='head_fused', core=core, case=name, us=round(t * 1e6, 1),
                        TFLOPs=round(flops / t / 1e12, 2), frac_mfma=round(flops / t / peak, 3),
                        GBps=round(nbytes / t / 1e9, 1), frac_hbm=round(nbytes / t / HBM, 3)))
    return out


def bench_warp_pyramid():
    from metrabs_amd.multiperson.multiperson_model import tta_parameters
    out = []
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (8, 3, 1080, 1920), dtype=torch.uint8, generator=g).cuda()
    for name, mat, l0b in [('8 x 1080p, f32 level 0 materialised', True, 4), ('8 x 1080p, uint8 level 0 (default)', False, 0)]:
        t = timeit(lambda: kernels.build_pyramid(frames, materialize_level0=mat))
        nbytes = 8 * 3 * (1080 * 1920 * (1 + l0b) + 540 * 960 * 4 + 270 * 480 * 4)
        out.append(dict(kernel='pyramid', case=name, us=round(t * 1e6, 1),
                        GBps=round(nbytes / t / 1e9, 1), frac_hbm=round(nbytes / t / HBM, 3)))
    frames_il = frames.contiguous(memory_format=torch.channels_last)   # [N,H,W,3] memory, sampled in place
    t = timeit(lambda: kernels.build_pyramid(frames_il))
    nbytes = 8 * 3 * (1080 * 1920 + 540 * 960 * 4 + 270 * 480 * 4)
    out.append(dict(kernel='pyramid', case='8 x 1080p, uint8 level 0, interleaved frames', us=round(t * 1e6, 1),
                    GBps=round(nbytes / t / 1e9, 1), frac_hbm=round(nbytes / t / HBM, 3)))
    pyr = kernels.build_pyramid(frames)
    pyr_il = kernels.build_pyramid(frames_il)
    for name, n, num_aug, aa, dt in [('64 crops 256px f32', 64, 1, 1, torch.float32),
                                     ('64 crops 256px f16', 64, 1, 1, torch.float16),
                                     ('320 crops (64x5 aug) f16', 64, 5, 1, torch.float16),
                                     ('64 crops aa=2 f32', 64, 1, 2, torch.float32)]:
        tta = {k: v.cuda() for k, v in tta_parameters(num_aug).items()}
        bw = 60 + 340 * torch.rand(n, generator=g)
        bh = 150 + 750 * torch.rand(n, generator=g)
        boxes = torch.stack([torch.rand(n, generator=g) * (1920 - bw),
                             torch.rand(n, generator=g) * (1080 - bh).clamp_min(1), bw, bh], 1).cuda()
        K = torch.tensor([[1844.0, 0, 960], [0, 1844.0, 540], [0, 0, 1]]).repeat(n, 1, 1).cuda()
        up = torch.tensor([0.0, -1, 0]).repeat(n, 1).cuda()
        ids = (torch.arange(n) % 8).int().cuda()
        dist = torch.zeros(n, 12).cuda()
        geo = lambda: kernels.crop_geometry(boxes, K, dist, up, ids,
                                            tta['rotflipmat'], tta['scales'], tta['gammas'], 256, aa)
        tg = timeit(geo)
        _, _, wp = geo()
        o = torch.empty(n * num_aug, 3, 256, 256, device='cuda', dtype=dt)
        t = timeit(lambda: kernels.warp_crops(pyr, wp, 256, aa, out=o))
        nbytes = o.numel() * o.element_size()
        # algorithmic bytes = the crops written + the clipped source footprint of every crop quad
        # (bench.source_footprint_bytes); the same 8 frames every launch, i.e. the
        # gathers are Infinity-Cache-assisted (bench.py times the 64-crop shape on rotating frames too)
        import bench
        src = bench.source_footprint_bytes(wp, 256 * aa, 1080, 1920)
        out.append(dict(kernel='warp', case=name, us=round(t * 1e6, 1), geometry_us=round(tg * 1e6, 1),
                        out_GBps=round(nbytes / t / 1e9, 1), crops_per_s=round(n * num_aug / t),
                        algorithmic_MB=round((nbytes + src) / 1e6, 1),
                        frac_hbm_cache_assisted=round((nbytes + src) / t / HBM, 3)))
        t_il = timeit(lambda: kernels.warp_crops(pyr_il, wp, 256, aa, out=o))
        out[-1]['interleaved_frames_us'] = round(t_il * 1e6, 1)
    return out


def bench_recon():
    out = []
    g = torch.Generator(device='cuda').manual_seed(0)
    for B, J in [(64, 17), (32, 122), (4096, 17)]:
        c2d = torch.rand(B, J, 2, device='cuda', generator=g) * 200 + 28
        rel = torch.randn(B, J, 3, device='cuda', generator=g) * 300
        K = torch.tensor([[500.0, 0, 128], [0, 500.0, 128], [0, 0, 1]], device='cuda').repeat(B, 1, 1)
        ws = kernels.reconstruct_workspace(B, J, 'cuda')
        o = torch.empty(B, J, 3, device='cuda')
        t = timeit(lambda: kernels.reconstruct_absolute(c2d, rel, K, MetrabsConfig(), workspace=ws, out=o))
        out.append(dict(kernel='reconstruct', case=f'B={B} J={J}', us=round(t * 1e6, 1)))
    return out


def bench_detector_pre():
    """K9: detector pre-processing (person_detector.py:21-33).  Algorithmic bytes: the uint8 frames
    once + the f32 network input once."""
    out = []
    g = torch.Generator().manual_seed(0)
    for name, n, h, w in [('8 x 1080p -> 256x416', 8, 1080, 1920), ('1 x 1080p -> 256x416', 1, 1080, 1920),
                          ('2 x 2160p -> 256x416', 2, 2160, 3840), ('8 x 480x640 -> 320x416', 8, 480, 640)]:
        frames = torch.randint(0, 256, (n, 3, h, w), dtype=torch.uint8, generator=g).cuda()
        geom = kernels.detector_geometry(h, w)
        o = torch.empty(n, 3, geom.out_h, geom.out_w, device='cuda')
        nbytes = frames.numel() + o.numel() * 4
        for kern in ('stream', 'tile'):  # ('auto' = the streaming kernel on these tensors)
            t = timeit(lambda: kernels.detector_preprocess(frames, geom=geom, out=o, kernel=kern))
            out.append(dict(kernel='detector_pre', variant=kern, case=name, us=round(t * 1e6, 1),
                            GBps=round(nbytes / t / 1e9, 1), frac_hbm=round(nbytes / t / HBM, 3)))
    return out


def bench_filter():
    """K8: plausibility filter + pose NMS, one launch per call; beside it the oracle's torch-op
    formulation (what the reference runs) on the host."""
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import cases, cpu_ref
    out = []
    for name, reps in [('coco17_aug5', 8), ('crowd_aug2', 8)]:
        c = cases.filter_case(name)
        boxes, p3, p2 = c['boxes'] * reps, c['poses3d'] * reps, c['poses2d'] * reps
        counts = [len(b) for b in boxes]
        P3, P2, BX = torch.cat(p3).cuda(), torch.cat(p2).cuda(), torch.cat(boxes).cuda()
        t = timeit(lambda: kernels.filter_poses(P3, P2, BX, counts, c['edges'], c['mean_bones']), graph=False)
        t0 = time.perf_counter()
        for _ in range(3):
            cpu_ref.filter_poses(boxes, p3, p2, c['edges'], c['mean_bones'])
        tc = (time.perf_counter() - t0) / 3
        out.append(dict(kernel='pose_filter', case=f'{name} x{reps}: {len(counts)} images, {sum(counts)} poses',
                        us=round(t * 1e6, 1), cpu_torch_ops_us=round(tc * 1e6, 1)))
    return out


def bench_bias_act():
    """K10: in-place bias + activation on an activation tensor (read + write = the algorithmic bytes),
    beside the two torch kernels it replaces."""
    import torch.nn.functional as F
    out = []
    g = torch.Generator(device='cuda').manual_seed(0)
    for name, shape, dt in [('B=64 96ch 64x64 f32 silu', (64, 96, 64, 64), torch.float32),
                            ('B=64 24ch 128x128 f32 silu', (64, 24, 128, 128), torch.float32),
                            ('B=64 1536ch 8x8 f32 silu', (64, 1536, 8, 8), torch.float32),
                            ('B=64 96ch 64x64 f16 silu', (64, 96, 64, 64), torch.float16)]:
        y = torch.randn(shape, device='cuda', generator=g).to(dt)
        b = torch.randn(shape[1], device='cuda', generator=g)
        t = timeit(lambda: kernels.bias_act_(y, b, 'silu'))
        bb = b.view(1, -1, 1, 1).to(dt)
        tt = timeit(lambda: F.silu(y + bb))
        nbytes = 2 * y.numel() * y.element_size()
        out.append(dict(kernel='bias_act', case=name, us=round(t * 1e6, 1), torch_two_kernels_us=round(tt * 1e6, 1),
                        GBps=round(nbytes / t / 1e9, 1), frac_hbm=round(nbytes / t / HBM, 3)))
    return out


def bench_depthwise():
    """K11: depthwise 3x3 + bias + SiLU (+ plane mean) in one pass (read x + write y = the
    algorithmic bytes), beside PyTorch's depthwise kernel + the elementwise ops."""
    import torch.nn.functional as F
    out = []
    g = torch.Generator(device='cuda').manual_seed(0)
    for name, (B, C, H), stride, dt in [('B=64 960ch 16x16 s1 f32', (64, 960, 16), 1, torch.float32),
                                        ('B=64 1536ch 8x8 s1 f32', (64, 1536, 8), 1, torch.float32),
                                        ('B=64 256ch 32x32 s2 f32', (64, 256, 32), 2, torch.float32),
                                        ('B=64 960ch 16x16 s1 f16', (64, 960, 16), 1, torch.float16)]:
        x = torch.randn(B, C, H, H, device='cuda', generator=g).to(dt)
        w = torch.randn(C, 1, 3, 3, device='cuda', generator=g) * 0.3
        b = torch.randn(C, device='cuda', generator=g)
        t = timeit(lambda: kernels.depthwise3x3_bias_act(x, w, b, 'silu', stride, 1, want_mean=True))
        wd, bd = w.to(dt), b.to(dt)

        def torch_path():
            prev = torch.backends.cudnn.enabled
            torch.backends.cudnn.enabled = False
            try:
                y = F.silu(F.conv2d(x, wd, bd, stride, 1, groups=C))
            finally:
                torch.backends.cudnn.enabled = prev
            return y, y.mean((2, 3))
        tt = timeit(torch_path)
        oh = (H + 2 - 3) // stride + 1
        nbytes = (x.numel() + B * C * oh * oh) * x.element_size()
        out.append(dict(kernel='depthwise3x3', case=name, us=round(t * 1e6, 1), torch_ops_us=round(tt * 1e6, 1),
                        GBps=round(nbytes / t / 1e9, 1), frac_hbm=round(nbytes / t / HBM, 3)))
    return out


if __name__ == '__main__':
    which = sys.argv[1:] or ['decode', 'head', 'warp', 'recon', 'detector', 'filter', 'bias_act', 'depthwise']
    benches = [('decode', bench_decode), ('head', bench_head), ('warp', bench_warp_pyramid),
               ('recon', bench_recon), ('detector', bench_detector_pre), ('filter', bench_filter),
               ('bias_act', bench_bias_act), ('depthwise', bench_depthwise)]
    for name, fn in benches:
        if name in which or (name == 'warp' and 'pyramid' in which):
            for r in fn():
                print(json.dumps(r), flush=True)
