"""Cycle budget of the two-halves 16-bit head (csrc/head_pp.hip) from s_memtime stamps inside the kernel: a developer
build with -DMTR_PP_TRACE=1 (tools/experiments/variant_lib.py pptrace head_pp.hip -DMTR_PP_TRACE=1) sums, per wave, the
cycles of every part of a stage -- work (reads + MFMAs, or copy issue), wait for its copies, barrier, for both phases --
and the workgroups with blockIdx % 293 == 0 write theirs behind the launch's coords2d.  configs[4]'s shape at 32 and 256
crops.  One JSON line per traced workgroup -> stdout (the form of profiles/r05q_head16_cycle_trace.jsonl)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metrabs_amd import _lib  # noqa: E402
_lib.load(os.environ.get('MTR_PROBE_LIB') or os.path.join(ROOT, 'tools', 'experiments', '_build', 'libmtr_pptrace.so'))
from metrabs_amd import kernels  # noqa: E402
from metrabs_amd.config import MetrabsConfig  # noqa: E402

NAMES = ['phase0_work', 'phase0_wait_copies', 'phase0_barrier', 'phase1_work', 'phase1_wait_copies', 'phase1_barrier',
         'prologue', 'epilogue']


def main():
    C, J, D, H, W = 1280, 122, 8, 12, 12
    g = torch.Generator(device='cuda').manual_seed(3)
    cfg = MetrabsConfig(depth=D, proc_side=384)
    w = torch.randn(J * 9, C, device='cuda', generator=g) * 0.02
    b = torch.randn(J * 9, device='cuda', generator=g) * 0.1
    packed = kernels.head_pack_weights(w, b, J, D, torch.float16)
    for B in (32, 256):
        for nhwc in (False, True):
            feat = torch.randn(B, C, H, W, device='cuda', generator=g).half()
            if nhwc:
                feat = feat.contiguous(memory_format=torch.channels_last)
            n_wg = ((B + 7) // 8) * 8 * 5
            traced = (n_wg + 292) // 293
            c2d = torch.zeros(B * J * 2 + traced * 128, device='cuda')
            c3d = torch.empty(B, J, 3, device='cuda')
            for _ in range(3):
                kernels.head_fused(feat, packed, C, J, cfg, out=(c2d, c3d), dma_staging=6)
            torch.cuda.synchronize()
            t = c2d[B * J * 2:].cpu().reshape(traced, 8, 16)
            for k in range(traced):
                n_st = int(t[k, 0, 10])
                if n_st == 0:
                    continue
                row = dict(B=B, nhwc=nhwc, block=int(t[k, 0, 9]), stages=n_st)
                for half, ws in (('X', range(0, 4)), ('Y', range(4, 8))):
                    per_stage = {NAMES[i]: round(float(t[k, list(ws), i].mean()) / n_st, 1) for i in range(6)}
                    row[half] = dict(per_stage_cycles=per_stage, stage_total=round(sum(per_stage.values()), 1),
                                     prologue=round(float(t[k, list(ws), 6].mean())), epilogue=round(float(t[k, list(ws), 7].mean())),
                                     kernel_total=round(float(t[k, list(ws), 8].mean())))
                print(json.dumps(row), flush=True)


if __name__ == '__main__':
    with torch.inference_mode():
        main()
