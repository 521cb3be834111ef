"""Random-shape stress of the 16-bit fused head vs the oracle (developer probe; the bounded cases
live in tests/test_gpu_head.py).  Draws B, C (multiple of 8), J, D, H, W, dtype, layout, GPW."""
import os, sys, random, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(seed0, n, gpw):
    import torch
    from metrabs_amd import kernels
    from metrabs_amd.config import MetrabsConfig
    from oracle import cases, cpu_ref
    worst = 0.0
    rnd = random.Random(seed0)
    for it in range(n):
        H = rnd.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16]); W = rnd.choice([2, 4, 6, 8, 10, 12, 16])
        if (H * W) % 4 or H * W > 256:
            continue
        D = rnd.choice([1, 2, 4, 8, 8, 8, 15, 31, 63]); J = rnd.choice([1, 2, 5, 7, 17, 24, 33, 122])
        C = rnd.choice([8 * rnd.randint(1, 40), 64 * rnd.randint(1, 6)]); B = rnd.choice([1, 2, 3, 8, 9, 17, 40])
        dt = rnd.choice([torch.float16, torch.bfloat16]); nhwc = rnd.random() < 0.5
        if J * (1 + D) * H * W * B > 3e7:
            continue
        cfg = cpu_ref.HeadConfig(depth=D, proc_side=max(H, W) * 8, stride_test=8, stride_train=8)
        g = cases.gen(seed0 * 1000 + it)
        feat = torch.randn(B, C, H, W, generator=g).to(dt)
        w, b = cases.default_conv_init(J * (1 + D), C, g)
        w, b = w * 3, b * 3
        with torch.inference_mode():
            o2, o3 = cpu_ref.heads_forward(feat.float(), cases.head_weights_as_consumed(w, dt), b, J, cfg)
        packed = kernels.head_pack_weights(w.cuda(), b.cuda(), J, D, dt)
        f = feat.cuda()
        if nhwc:
            f = f.contiguous(memory_format=torch.channels_last)
        c2, c3 = kernels.head_fused(f, packed, C, J, MetrabsConfig.from_any(cfg.as_dict()), groups_per_workgroup=int(os.environ.get('HEAD_GPW', '0')))
        e3, e2 = float((c3.cpu() - o3).abs().max()), float((c2.cpu() - o2).abs().max())
        worst = max(worst, e3)
        if not (e3 <= 2e-3 and e2 <= 4e-4) or not torch.isfinite(c3).all():
            print('FAIL', dict(B=B, C=C, J=J, D=D, H=H, W=W, dt=str(dt), nhwc=nhwc, gpw=gpw), e3, e2, flush=True)
    print(f'gpw={gpw} seed={seed0}: {n} draws, worst coords3d error {worst:.2e} mm', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run(int(sys.argv[1]), int(sys.argv[2]), os.environ.get('HEAD_GPW', '0'))
    else:
        for gpw in ('0', '1', '2', '3'):
            subprocess.run([sys.executable, os.path.abspath(__file__), str(11 + int(gpw)), '150'],
                           env=dict(os.environ, HEAD_GPW=gpw))
