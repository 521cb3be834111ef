"""HIP graphs behind Pose3dEstimator's API: shape-bucketed capture of whole internal batches.

One internal batch (multiperson_model.py:189-220) is ~10 launches of ours + ~400 of the backbone.  At the
reference's default of 64 crops the host issues them faster than the GPU runs them (back-to-back calls
run at the GPU's rate either way, measured: DESIGN.md section 9); with fewer boxes per call the eager
path is bound by the host (37 crops per call: +18 % with graphs), and a single synchronous call always
pays the issue time.  A captured batch is ONE graph launch.  What is captured is exactly the call sequence
the eager path runs (``Pose3dEstimator._batch_with_postprocess``: crop geometry -> sampler -> crop model -> K7)
on exactly the same shapes, so a replay returns the eager path's bits.

* ``FrameSet`` -- static uint8 frames + their pyramid for one frame size (H, W), with room for the
  largest number of frames seen: the fixed addresses a graph's sampler reads (a call with fewer frames
  uses the head of the buffers); frames with interleaved channels (channels_last) get a set of their own
  and stay interleaved.  A call copies its frames in (host frames: the H2D copy lands there
  directly) and rebuilds the pyramid with one eager launch.
* ``BatchGraph`` -- one internal batch of n boxes captured against a FrameSet; static copies of the six
  per-box parameter arrays, one ``hipGraphLaunch`` per replay, the result cloned out.
* ``GraphCache`` -- the estimator's cache: key = (frames shape, n boxes, num_aug, antialias factor,
  crop dtype / layout, average_aug, skeleton, joint transform); a key is captured on its 2nd occurrence
  ('auto') or its first (True); LRU-bounded; ragged tails and one-off shapes stay eager.

Everything is ordered on the caller's current stream; one estimator serves one stream at a time.
Weights are read at capture.  A graph keeps the head's derived weight tensors it read alive and is
re-captured when they, the head's parameters or the crop model's storage (``.to`` / ``.half`` on it)
changed; in-place edits of BACKBONE parameters are the one case left to ``estimator.graphs.clear()``.
"""
import collections
import warnings

import torch

from metrabs_amd import kernels
from metrabs_amd.pipeline import CAPTURE_ERROR_MODE


class FrameSet:
    def __init__(self, n, h, w, device, hwc=False):
        self.key = (h, w, str(device), bool(hwc))
        self.capacity = n
        self.hwc = bool(hwc)   # frames kept (and sampled) with interleaved channels, as the caller's are
        # (buffers that outlive the call and are written in place by later ones: made OUTSIDE inference mode,
        #  or a first call under torch.inference_mode() would leave inference tensors that a later call under
        #  plain no_grad may not update)
        with torch.inference_mode(False):
            self.images = (torch.empty(n, h, w, 3, dtype=torch.uint8, device=device).permute(0, 3, 1, 2) if hwc
                           else torch.empty(n, 3, h, w, dtype=torch.uint8, device=device))
            _, self._l1, self._l2 = kernels._alloc_levels(n, h, w, device, with_level0=False)
            self._lut = torch.empty(256, device=device, dtype=torch.float32)
        self._views = {}
        self.pyramid = self.pyramid_of(n)
        self._copy_stream = None   # pinned host frames: H2D on a stream of its own, two staging buffers
        self._staging, self._staging_free, self._turn = None, None, 0

    def pyramid_of(self, n):
        """The pyramid over the first n frames of the buffers (same base addresses for every n: the levels
        are frame-major)."""
        if n not in self._views:
            self._views[n] = kernels.Pyramid([None, self._l1[:n], self._l2[:n]], images_u8=self.images[:n],
                                             lut=self._lut, hwc=self.hwc)
        return self._views[n]

    def load(self, images):
        """frames (host or device, uint8 [n,3,H,W], n <= capacity) -> the static pyramid of the first n
        frames (stream-ordered).
        Frames in PINNED host memory are copied on a copy stream into one of two staging buffers in HBM
        and from there (device to device, 50 MB at 1080p x 8: ~25 us) into the static frames: the PCIe
        copy of call i + 1 runs under the compute of call i, which is still queued on the caller's stream.
        The host waits for its own copy to finish (the caller's buffer is free again on return, as after a
        blocking ``.cuda()``); it does not wait for the GPU's compute."""
        if images.dtype != torch.uint8:
            raise ValueError('images must be uint8 [N,3,H,W]')
        n = len(images)
        if n > self.capacity:
            raise ValueError(f'{n} frames for a frame set of {self.capacity}')
        pyramid = self.pyramid_of(n)
        dst = pyramid.images_u8
        if not images.is_cuda and images.is_pinned():
            dev = self.images.device
            cur = torch.cuda.current_stream(dev)
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(dev)
                with torch.inference_mode(False):
                    self._staging = [torch.empty_like(self.images) for _ in range(2)]
                self._staging_free = [torch.cuda.Event() for _ in range(2)]
                # the allocator may have handed out blocks that kernels still queued on the caller's
                # stream (the previous call's -- the host runs ahead by design) read or write: the first
                # copies into them wait for everything queued so far
                self._copy_stream.wait_stream(cur)
            self._turn ^= 1
            b = self._turn
            ready = torch.cuda.Event()
            with torch.cuda.stream(self._copy_stream):
                self._copy_stream.wait_event(self._staging_free[b])  # the D2D copy that last read this buffer
                self._staging[b][:n].copy_(images, non_blocking=True)
                ready.record(self._copy_stream)
            cur.wait_event(ready)
            dst.copy_(self._staging[b][:n], non_blocking=True)
            self._staging_free[b].record(cur)
            ready.synchronize()
        else:
            dst.copy_(images, non_blocking=True)
        return kernels.build_pyramid(dst, out=pyramid)


def _last_parameter(module):
    """The parameter `list(module.parameters())[-1]` would give, found along the tail of the module tree
    only (a module's own parameters precede its children's)."""
    for child in reversed(list(module._modules.values())):
        if child is not None:
            p = _last_parameter(child)
            if p is not None:
                return p
    own = [p for p in module._parameters.values() if p is not None]
    return own[-1] if own else None


def backbone_fingerprint(crop_model):
    """Cheap identity of the storage a captured graph reads the BACKBONE's weights from: the module object and
    the address / dtype / device of its first and last parameter.  `.half()`, `.to()`, `.cuda()` on the backbone
    alone, or a replaced backbone, change it (Metrabs._apply only sees calls on the whole crop model; ADVICE r5:
    a replay would otherwise read freed storage).  In-place edits of backbone parameters keep the addresses and
    are, as documented, the caller's to announce (`estimator.graphs.clear()`)."""
    bb = getattr(crop_model, 'backbone', None)
    if not isinstance(bb, torch.nn.Module):
        return (id(bb),)
    edge = (next(bb.parameters(), None), _last_parameter(bb))
    return (id(bb),) + tuple(None if p is None else (p.data_ptr(), p.dtype, str(p.device)) for p in edge)


class BatchGraph:
    """One internal batch captured in a HIP graph.  ``replay(batch_args)`` -> [n, (A,) S, 5]."""

    def __init__(self, est, frames, batch_args, tta, antialias_factor, post, warmup=2, n_frames=None,
                 pool=None):
        self.frames = frames
        pyramid = frames.pyramid if n_frames is None else frames.pyramid_of(n_frames)
        self.heads = getattr(est.crop_model, 'heatmap_heads', None)
        self.storage_generation = getattr(est.crop_model, 'storage_generation', None)
        self.backbone = backbone_fingerprint(est.crop_model)
        with torch.inference_mode(False):  # (written in place by every later replay, whatever mode it runs under)
            self.static = [torch.empty(a.shape, dtype=a.dtype, device=a.device) for a in batch_args]
        self._load(batch_args)
        body = lambda: est._batch_with_postprocess(pyramid, *self.static, tta, antialias_factor, post)
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
            kw = {} if pool is None else dict(pool=pool)
            with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_ERROR_MODE, **kw):
                self.out = body()
        # tensors outside the graph's pool that the captured kernels read by address: the head's derived
        # weights (packed tiles, the latent prefix's rows).  Holding them keeps the allocator from handing
        # the memory out again; is_current() says whether they are still what an eager call would use.
        self.external = self.heads.packed_snapshot() if hasattr(self.heads, 'packed_snapshot') else {}
        self.replays = 0

    def is_current(self, est):
        """False when the weights this graph read were replaced or edited since its capture (the head's
        parameters / derived tensors; the crop model, or its backbone alone, moved, cast or replaced)."""
        if getattr(est.crop_model, 'storage_generation', None) != self.storage_generation:
            return False
        if backbone_fingerprint(est.crop_model) != self.backbone:   # the backbone alone was cast / moved / replaced
            return False
        heads = getattr(est.crop_model, 'heatmap_heads', None)
        if heads is not self.heads:
            return False
        return not hasattr(heads, 'snapshot_is_current') or heads.snapshot_is_current(self.external)

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
    def __init__(self, estimator, max_graphs=64, max_frame_sets=4, min_batches_between_evictions=128):
        self.est = estimator
        # (64 since round 5: every box count of an internal batch of up to 64 crops fits; the captured batches share
        #  one allocator pool, so a graph costs its own static parameters and output, not a set of activations)
        self.max_graphs = max_graphs
        self.max_frame_sets = max_frame_sets
        # A capture costs a few eager batches (warm-up + the capture itself).  While the cache has room a
        # shape is captured on its 2nd (True: 1st) occurrence; once it is FULL, a new shape may push out the
        # least recently used graph only every `min_batches_between_evictions` batches -- a server whose box
        # count wanders over more shapes than the cache holds keeps replaying what it has and runs the rest
        # eagerly instead of capturing on every call.  The same limit holds for frame sets: replacing one
        # (another frame size when all sets are in use, or more frames than a set has room for) drops
        # every graph that reads it, so a caller cycling over more frame sizes than `max_frame_sets` runs
        # the sizes that do not fit eagerly instead of rebuilding buffers and graphs on every call.
        self.min_batches_between_evictions = min_batches_between_evictions
        self._batches = 0
        self._last_eviction_at = -(1 << 60)
        self.graphs = collections.OrderedDict()
        self.frame_sets = collections.OrderedDict()
        self.seen = collections.Counter()
        self.failed = set()
        self.stats = dict(captures=0, replays=0, eager_batches=0, evictions=0, stale=0)
        self.last_capture_error = None
        self._pool = None   # one allocator pool shared by every captured batch (they never run concurrently)

    def clear(self):
        self.graphs.clear()
        self.frame_sets.clear()
        self.seen.clear()
        self.failed.clear()

    def _threshold(self):
        mode = self.est.graph_batches
        return 1 if mode is True else 2

    def _eviction_allowed(self):
        return self._batches - self._last_eviction_at >= self.min_batches_between_evictions

    def _set_needs_eviction(self, n, skey):
        """Would serving n frames of this size replace a frame set that graphs may read?"""
        fs = self.frame_sets.get(skey)
        if fs is not None:
            return fs.capacity < n
        return len(self.frame_sets) >= self.max_frame_sets

    def plan_call(self, images, ranges, tta, antialias_factor, post):
        """-> _CallPlan when at least one internal batch of this call has, or is now due, a graph; else
        None (the call runs as before: no frame copy, no static buffers)."""
        if torch.cuda.is_current_stream_capturing():
            return None  # (the caller is capturing the whole call itself)
        est = self.est
        dev = est._device()
        n, _, h, w = images.shape
        hwc = kernels.frames_are_interleaved(images)
        skey = (h, w, str(dev), hwc)
        fkey = (n,) + skey
        jt = post['joint_transform']
        base = (fkey, len(tta['gammas']), int(antialias_factor), est.crop_dtype, bool(est.crop_channels_last),
                bool(post['average_aug']), post['skeleton'].data_ptr(), None if jt is None else jt.data_ptr(),
                int(est.crop_model.input_resolution))
        keys = [base + (stop - start,) for start, stop in ranges]
        threshold = self._threshold()
        # a frame set would have to go (with its graphs): allowed once per eviction interval, and decided
        # ONCE per call -- the batches of one call share the frame set
        set_blocked = self._set_needs_eviction(n, skey) and not self._eviction_allowed()
        use = []
        for k, (start, stop) in zip(keys, ranges):
            if stop == start or k in self.failed:
                use.append(False)
                continue
            self.seen[k] += 1
            self._batches += 1
            g = self.graphs.get(k)
            if g is not None and not g.is_current(est):   # its weights were replaced: capture again
                del self.graphs[k]
                self.stats['stale'] += 1
                g = None
            if set_blocked:
                use.append(False)
                continue
            due = self.seen[k] >= threshold
            if due and g is None and len(self.graphs) >= self.max_graphs:
                due = self._eviction_allowed()
                if due:
                    self._last_eviction_at = self._batches
            use.append(g is not None or due)
        if len(self.seen) > 4096:
            self.seen.clear()
        if not any(use):
            self.stats['eager_batches'] += len(use)
            return None
        frames = self.frame_set(n, h, w, dev, hwc=hwc)
        return _CallPlan(self, frames, keys, use, tta, antialias_factor, post)

    def frame_set(self, n, h, w, dev, optional=False, hwc=False):
        """The static frame + pyramid buffers for n frames of this size (one set per frame size, grown to the
        largest n seen; LRU over max_frame_sets; the graphs captured against a replaced set go with it).
        optional=True (the pinned-frame staging of an eager call): None instead of a replacement the
        eviction interval does not allow yet.  hwc: frames with interleaved channels (a set of its own)."""
        skey = (h, w, str(dev), bool(hwc))
        if optional:
            # an eager call's staging counts towards the eviction interval too: with graph_batches off nothing
            # else advances the counter and the first replacement would otherwise block every later one
            self._batches += 1
        frames = self.frame_sets.get(skey)
        if frames is not None and frames.capacity >= n:
            self.frame_sets.move_to_end(skey)
            return frames
        if self._set_needs_eviction(n, skey):
            if optional and not self._eviction_allowed():
                return None
            if optional:
                # the set that would go: this size's own (too small) or the least recently used one.  Graphs that
                # read it would go with it -- for a call that replays none of them: take the plain upload instead
                victim = frames if frames is not None else self.frame_sets[next(iter(self.frame_sets))]
                if any(g.frames is victim for g in self.graphs.values()):
                    return None
            self._last_eviction_at = self._batches
        if frames is not None:       # more frames than the set has room for: a larger one takes its place
            self._drop_frame_set(skey)
        frames = FrameSet(n, h, w, dev, hwc=hwc)
        self.frame_sets[skey] = frames
        while len(self.frame_sets) > self.max_frame_sets:
            self._drop_frame_set(next(iter(self.frame_sets)))
        return frames

    def _drop_frame_set(self, skey):
        old = self.frame_sets.pop(skey)
        for k in [k for k, g in self.graphs.items() if g.frames is old]:  # they read the dropped buffers
            del self.graphs[k]
            self.seen[k] = 0     # a shape that comes back earns its graph again
            self.stats['evictions'] += 1

    def _get_or_capture(self, key, frames, batch_args, tta, antialias_factor, post):
        g = self.graphs.get(key)
        if g is not None and g.frames is frames:
            self.graphs.move_to_end(key)
            self.stats['replays'] += 1
            return g
        try:
            if self._pool is None and torch.cuda.is_available():
                self._pool = torch.cuda.graph_pool_handle()
            g = BatchGraph(self.est, frames, batch_args, tta, antialias_factor, post, n_frames=key[0][0],
                           pool=self._pool)
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
