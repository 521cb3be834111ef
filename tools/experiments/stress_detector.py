"""Stress of the detector pre-processing: the streaming kernel against the tile kernel on many random
launches (sizes that give several units per workgroup, ring wrap-arounds, partial last chunks), and
repeated launches of one input for run-to-run identity.  Developer tool (run on the GPU box)."""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrabs_amd import kernels  # noqa: E402

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
g = torch.Generator(device='cuda').manual_seed(7)
bad = 0
n_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for it in range(n_launch):
    n = rng.randint(1, 12)
    h = rng.randint(40, 1500)
    w = 16 * rng.randint(3, 130)
    frames = torch.randint(0, 256, (n, 3, h, w), dtype=torch.uint8, device='cuda', generator=g)
    try:
        tile, geom = kernels.detector_preprocess(frames, kernel='tile')
    except RuntimeError:
        continue
    stream, _ = kernels.detector_preprocess(frames, kernel='stream')
    if not torch.equal(tile, stream):
        bad += 1
        print('MISMATCH', (n, h, w), float((tile - stream).abs().max()), flush=True)
    if it % 25 == 0:  # the same launch five times: identical bits every time
        for _ in range(5):
            again, _ = kernels.detector_preprocess(frames, kernel='stream')
            if not torch.equal(again, stream):
                bad += 1
                print('NOT REPEATABLE', (n, h, w), flush=True)
print(f'{n_launch} launches, {bad} mismatches')
sys.exit(1 if bad else 0)
