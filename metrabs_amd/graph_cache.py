"""HIP graphs behind Pose3dEstimator's API: shape-bucketed capture of whole internal batches.

One internal batch (multiperson_model.py:189-220) is ~10 launches of ours + ~400 of the backbone.  At the
reference's default of 64 crops the host issues them faster than the GPU runs them (back-to-back calls
run at the GPU's rate either way, measured: DESIGN.md section 9); with fewer boxes per call the eager
path is bound by the host (37 crops per call: +18 % with graphs), and a single synchronous call always
pays the issue time.  A captured batch is ONE graph launch.  What is captured is exactly the call sequence
the eager path runs (``Pose3dEstimator._batch_with_postprocess``: crop geometry -> sampler -> crop model -> K7)
on exactly the same shapes, so a replay returns the eager path's bits.

* ``FrameSet`` -- static uint8 frames + their pyramid for one (n_frames, H, W): the fixed addresses a
  graph's sampler reads.  A call copies its frames in (host frames: the H2D copy lands there
  directly) and rebuilds the pyramid with one eager launch.
* ``BatchGraph`` -- one internal batch of n boxes captured against a FrameSet; static copies of the six
  per-box parameter arrays, one ``hipGraphLaunch`` per replay, the result cloned out.
* ``GraphCache`` -- the estimator's cache: key = (frames shape, n boxes, num_aug, antialias factor,
  crop dtype / layout, average_aug, skeleton, joint transform); a key is captured on its 2nd occurrence
  ('auto') or its first (True); LRU-bounded; ragged tails and one-off shapes stay eager.

Everything is ordered on the caller's current stream; one estimator serves one stream at a time.
Weights are read at capture: after changing them call ``estimator.graphs.clear()``.
"""
import collections
import warnings

import torch

from metrabs_amd import kernels
from metrabs_amd.pipeline import CAPTURE_ERROR_MODE


class FrameSet:
    def __init__(self, n, h, w, device):
        self.key = (n, h, w, str(device))
        # (buffers that outlive the call and are written in place by later ones: made OUTSIDE inference mode,
        #  or a first call under torch.inference_mode() would leave inference tensors that a later call under
        #  plain no_grad may not update)
        with torch.inference_mode(False):
            self.images = torch.empty(n, 3, h, w, dtype=torch.uint8, device=device)
            _, l1, l2 = kernels._alloc_levels(n, h, w, device, with_level0=False)
            lut = torch.empty(256, device=device, dtype=torch.float32)
        self.pyramid = kernels.Pyramid([None, l1, l2], images_u8=self.images, lut=lut)
        self._copy_stream = None   # pinned host frames: H2D on a stream of its own, two staging buffers
        self._staging, self._staging_free, self._turn = None, None, 0

    def load(self, images):
        """frames (host or device, uint8 [n,3,H,W]) -> the static pyramid (stream-ordered).
        Frames in PINNED host memory are copied on a copy stream into one of two staging buffers in HBM
        and from there (device to device, 50 MB at 1080p x 8: ~25 us) into the static frames: the PCIe
        copy of call i + 1 runs under the compute of call i, which is still queued on the caller's stream.
        The host waits for its own copy to finish (the caller's buffer is free again on return, as after a
        blocking ``.cuda()``); it does not wait for the GPU's compute."""
        if images.dtype != torch.uint8:
            raise ValueError('images must be uint8 [N,3,H,W]')
        if not images.is_cuda and images.is_pinned():
            dev = self.images.device
            cur = torch.cuda.current_stream(dev)
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(dev)
                with torch.inference_mode(False):
                    self._staging = [torch.empty_like(self.images) for _ in range(2)]
                self._staging_free = [torch.cuda.Event() for _ in range(2)]
            self._turn ^= 1
            b = self._turn
            ready = torch.cuda.Event()
            with torch.cuda.stream(self._copy_stream):
                self._copy_stream.wait_event(self._staging_free[b])  # the D2D copy that last read this buffer
                self._staging[b].copy_(images, non_blocking=True)
                ready.record(self._copy_stream)
            cur.wait_event(ready)
            self.images.copy_(self._staging[b], non_blocking=True)
            self._staging_free[b].record(cur)
            ready.synchronize()
        else:
            self.images.copy_(images, non_blocking=True)
        return kernels.build_pyramid(self.images, out=self.pyramid)


class BatchGraph:
    """One internal batch captured in a HIP graph.  ``replay(batch_args)`` -> [n, (A,) S, 5]."""

    def __init__(self, est, frames, batch_args, tta, antialias_factor, post, warmup=2):
        self.frames = frames
        with torch.inference_mode(False):  # (written in place by every later replay, whatever mode it runs under)
            self.static = [torch.empty(a.shape, dtype=a.dtype, device=a.device) for a in batch_args]
        self._load(batch_args)
        body = lambda: est._batch_with_postprocess(frames.pyramid, *self.static, tta, antialias_factor, post)
        # ALWAYS captured under inference mode, whatever the caller runs under: torch creates the CUDA
        # generator's graph-state tensors at the process's first capture and updates them in place at every
        # later capture_begin -- a first capture under inference_mode (ours, the bench's pipeline, a test's)
        # followed by one under plain no_grad fails with "inplace update to inference tensor"; in-place updates
        # of normal tensors under inference mode are fine, so this order-independent rule works both ways.
        # (The captured output is cloned on every replay: callers get normal tensors outside inference mode.)
        with torch.inference_mode():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):  # lazy initialisation (MIOpen's solver search, weight packing)
                    body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_ERROR_MODE):
                self.out = body()
        self.replays = 0

    def _load(self, batch_args):
        for dst, src in zip(self.static, batch_args):
            dst.copy_(src, non_blocking=True)

    def replay(self, batch_args):
        self._load(batch_args)
        self.graph.replay()
        self.replays += 1
        return self.out.clone()


class _CallPlan:
    """What GraphCache.plan_call hands to _predict_in_batches for ONE call: the frame set and, per
    internal batch, whether it replays / captures a graph or runs eagerly."""

    def __init__(self, cache, frames, keys, use, tta, antialias_factor, post):
        self.cache, self.frames, self.keys, self.use = cache, frames, keys, use
        self.tta, self.aa, self.post = tta, antialias_factor, post

    def graph_for(self, i_range, batch_args):
        if not self.use[i_range]:
            self.cache.stats['eager_batches'] += 1
            return None
        return self.cache._get_or_capture(self.keys[i_range], self.frames, batch_args, self.tta, self.aa,
                                          self.post)


class GraphCache:
    def __init__(self, estimator, max_graphs=32, max_frame_sets=2, min_batches_between_evictions=128):
        self.est = estimator
        self.max_graphs = max_graphs
        self.max_frame_sets = max_frame_sets
        # A capture costs a few eager batches (warm-up + the capture itself).  While the cache has room a
        # shape is captured on its 2nd (True: 1st) occurrence; once it is FULL, a new shape may push out the
        # least recently used graph only every `min_batches_between_evictions` batches -- a server whose box
        # count wanders over more shapes than the cache holds keeps replaying what it has and runs the rest
        # eagerly instead of capturing on every call.
        self.min_batches_between_evictions = min_batches_between_evictions
        self._batches = 0
        self._last_eviction_at = -(1 << 60)
        self.graphs = collections.OrderedDict()
        self.frame_sets = collections.OrderedDict()
        self.seen = collections.Counter()
        self.failed = set()
        self.stats = dict(captures=0, replays=0, eager_batches=0, evictions=0)
        self.last_capture_error = None

    def clear(self):
        self.graphs.clear()
        self.frame_sets.clear()
        self.seen.clear()
        self.failed.clear()

    def _threshold(self):
        mode = self.est.graph_batches
        return 1 if mode is True else 2

    def plan_call(self, images, ranges, tta, antialias_factor, post):
        """-> _CallPlan when at least one internal batch of this call has, or is now due, a graph; else
        None (the call runs as before: no frame copy, no static buffers)."""
        if torch.cuda.is_current_stream_capturing():
            return None  # (the caller is capturing the whole call itself)
        est = self.est
        dev = est._device()
        n, _, h, w = images.shape
        fkey = (n, h, w, str(dev))
        jt = post['joint_transform']
        base = (fkey, len(tta['gammas']), int(antialias_factor), est.crop_dtype, bool(est.crop_channels_last),
                bool(post['average_aug']), post['skeleton'].data_ptr(), None if jt is None else jt.data_ptr(),
                int(est.crop_model.input_resolution))
        keys = [base + (stop - start,) for start, stop in ranges]
        threshold = self._threshold()
        use = []
        for k, (start, stop) in zip(keys, ranges):
            if stop == start or k in self.failed:
                use.append(False)
                continue
            self.seen[k] += 1
            self._batches += 1
            due = self.seen[k] >= threshold
            if due and k not in self.graphs and len(self.graphs) >= self.max_graphs:
                due = self._batches - self._last_eviction_at >= self.min_batches_between_evictions
                if due:
                    self._last_eviction_at = self._batches
            use.append(k in self.graphs or due)
        if len(self.seen) > 4096:
            self.seen.clear()
        if not any(use):
            self.stats['eager_batches'] += len(use)
            return None
        frames = self.frame_set(n, h, w, dev)
        return _CallPlan(self, frames, keys, use, tta, antialias_factor, post)

    def frame_set(self, n, h, w, dev):
        """The static frame + pyramid buffers for frames of this shape (LRU over max_frame_sets; the graphs
        captured against an evicted set go with it)."""
        fkey = (n, h, w, str(dev))
        frames = self.frame_sets.get(fkey)
        if frames is None:
            frames = FrameSet(n, h, w, dev)
            self.frame_sets[fkey] = frames
            while len(self.frame_sets) > self.max_frame_sets:
                old_key, _ = self.frame_sets.popitem(last=False)
                for k in [k for k in self.graphs if k[0] == old_key]:  # they read the evicted buffers
                    del self.graphs[k]
                    self.stats['evictions'] += 1
        else:
            self.frame_sets.move_to_end(fkey)
        return frames

    def _get_or_capture(self, key, frames, batch_args, tta, antialias_factor, post):
        g = self.graphs.get(key)
        if g is not None and g.frames is frames:
            self.graphs.move_to_end(key)
            self.stats['replays'] += 1
            return g
        try:
            g = BatchGraph(self.est, frames, batch_args, tta, antialias_factor, post)
        except Exception as e:  # noqa: BLE001 -- whatever a capture can raise: this shape stays eager
            import traceback
            self.failed.add(key)
            self.last_capture_error = traceback.format_exc()
            warnings.warn(f'metrabs_amd: HIP graph capture of an internal batch failed ({str(e)[:200]}); '
                          f'this shape keeps running eagerly (estimator.graphs.last_capture_error has the traceback)')
            torch.cuda.synchronize()
            self.stats['eager_batches'] += 1
            return None
        self.graphs[key] = g
        self.stats['captures'] += 1
        while len(self.graphs) > self.max_graphs:
            self.graphs.popitem(last=False)
            self.stats['evictions'] += 1
        return g
