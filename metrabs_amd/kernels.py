"""Tensor-level wrappers of the C-ABI (include/metrabs_hip.h): torch tensors in, torch tensors out,
kernels enqueued on torch's current HIP stream.  torch is used for device memory and streams only.
"""
import ctypes

import torch

from metrabs_amd import _lib
from metrabs_amd._lib import check, current_stream_ptr, dtype_code, require_cuda


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def softargmax_decode(logits, n_points, cfg, out=None, nhwc_staging=0):
    """logits [B, J*(1+D), H, W] (f32/f16/bf16, NCHW) -> (coords2d [B,J,2] px, coords3d_rel [B,J,3] mm).
    MetrabsHeads.forward after the conv (metrabs_pytorch/models/metrabs.py:78-85).
    nhwc_staging (channels_last logits; mtr_softargmax_decode_opts): 0 = the library's kernel choice, 1 = never the
    LDS-staged kernel, 2 = whenever the shape allows, 3 = as 2 with two crops per workgroup where that fills the waves; the same bits either way."""
    require_cuda(logits)
    lib = _lib.load()
    # torch channels_last logits (what a channels_last conv_final emits; the TF twin's
    # 'b h w (d j)' layout, metrabs_tf/models/metrabs.py:100-101) are decoded in place
    nhwc = _is_channels_last(logits)
    if not nhwc:
        logits = logits.contiguous()
    B, n_out, H, W = logits.shape
    J = int(n_points)
    D = n_out // J - 1
    if n_out != J * (1 + D) or D != cfg.depth:
        raise ValueError(f'logits have {n_out} channels, expected J*(1+depth) = {J * (1 + cfg.depth)}')
    if out is None:
        c2d = torch.empty(B, J, 2, device=logits.device, dtype=torch.float32)
        c3d = torch.empty(B, J, 3, device=logits.device, dtype=torch.float32)
    else:
        c2d, c3d = out
    hp = cfg.head_params()
    check(lib.mtr_softargmax_decode_opts(
        _ptr(logits), dtype_code(logits.dtype), _lib.MTR_NHWC if nhwc else _lib.MTR_NCHW, B, J, D, H, W,
        ctypes.byref(hp), int(nhwc_staging), _ptr(c2d), _ptr(c3d), current_stream_ptr(logits.device)),
        'mtr_softargmax_decode')
    return c2d, c3d


def reconstruct_workspace(B, J, device):
    n = _lib.load().mtr_reconstruct_workspace_bytes(int(B), int(J))
    return torch.empty((n + 7) // 8, device=device, dtype=torch.float64)


def reconstruct_absolute(coords2d, coords3d_rel, intrinsics, cfg, mix_3d_inside_fov='cfg',
                         weak_perspective=None, workspace=None, out=None):
    """ptu3d.reconstruct_absolute (metrabs_pytorch/ptu3d.py:9-33)."""
    require_cuda(coords2d, coords3d_rel, intrinsics)
    lib = _lib.load()
    coords2d = coords2d.contiguous().float()
    coords3d_rel = coords3d_rel.contiguous().float()
    intrinsics = intrinsics.contiguous().float()
    B, J = coords2d.shape[:2]
    if intrinsics.shape != (B, 3, 3) or coords3d_rel.shape != (B, J, 3):
        raise ValueError('shape mismatch between coords2d, coords3d_rel and intrinsics')
    if workspace is None:
        workspace = reconstruct_workspace(B, J, coords2d.device)
    if out is None:
        out = torch.empty(B, J, 3, device=coords2d.device, dtype=torch.float32)
    rp = cfg.recon_params(mix_3d_inside_fov, weak_perspective)
    check(lib.mtr_reconstruct_absolute(
        _ptr(coords2d), _ptr(coords3d_rel), _ptr(intrinsics), B, J, ctypes.byref(rp), _ptr(out),
        _ptr(workspace), workspace.numel() * 8, current_stream_ptr(coords2d.device)),
        'mtr_reconstruct_absolute')
    return out


def reconstruct_moments(coords2d, coords3d_rel, intrinsics, workspace=None):
    """-> f64 [3] = (sum normalized2d^2, sum rel_backproj^2, count) over this call's crops."""
    require_cuda(coords2d, coords3d_rel, intrinsics)
    lib = _lib.load()
    B, J = coords2d.shape[:2]
    if workspace is None:
        workspace = reconstruct_workspace(B, J, coords2d.device)
    moments = torch.zeros(3, device=coords2d.device, dtype=torch.float64)
    if B > 0:
        check(lib.mtr_reconstruct_moments(
            _ptr(coords2d), _ptr(coords3d_rel), _ptr(intrinsics), B, J, _ptr(moments),
            _ptr(workspace), workspace.numel() * 8, current_stream_ptr(coords2d.device)),
            'mtr_reconstruct_moments')
    return moments


def reconstruct_solve(coords2d, coords3d_rel, intrinsics, moments, cfg, mix_3d_inside_fov='cfg',
                      weak_perspective=None, out=None):
    require_cuda(coords2d, coords3d_rel, intrinsics, moments)
    lib = _lib.load()
    B, J = coords2d.shape[:2]
    if out is None:
        out = torch.empty(B, J, 3, device=coords2d.device, dtype=torch.float32)
    rp = cfg.recon_params(mix_3d_inside_fov, weak_perspective)
    check(lib.mtr_reconstruct_solve(
        _ptr(coords2d), _ptr(coords3d_rel), _ptr(intrinsics), B, J, ctypes.byref(rp),
        _ptr(moments), _ptr(out), current_stream_ptr(coords2d.device)), 'mtr_reconstruct_solve')
    return out


class Pyramid:
    """Linear-light image pyramid of the reference (multiperson_model.py:196 + warping.py:10-13).

    Two representations of level 0:
      * ``levels[0]`` f32 [N,3,H,W] (materialised -- what warping.warp_images_with_pyramid builds);
      * ``images_u8`` + ``lut``: level 0 stays the uint8 frame, decoded through the 256-entry gamma
        LUT inside the sampler (``levels[0]`` is None).  Same numbers, 64 % fewer pyramid bytes.
    ``levels[1]``, ``levels[2]`` are always f32 planes.  ``hwc``: ``images_u8`` is [N,3,H,W] over
    interleaved memory ([N,H,W,3]: torch channels_last, what decoders / numpy frames are) and is sampled
    as it lies -- the sampler then needs two gathers per sample instead of six."""

    def __init__(self, levels, images_u8=None, lut=None, hwc=False):
        self.levels = levels
        self.images_u8 = images_u8
        self.lut = lut
        self.hwc = bool(hwc)
        ref = images_u8 if levels[0] is None else levels[0]
        self.n, _, self.h, self.w = ref.shape
        self.device = ref.device


def frames_are_interleaved(images):
    """[N,3,H,W] frames whose memory is [N,H,W,3] (``x.permute(0, 3, 1, 2)`` of an HWC batch,
    ``memory_format=torch.channels_last``)?  Plain contiguous frames -- also the shapes where the two
    orders coincide -- are not."""
    return (images.ndim == 4 and images.shape[1] == 3 and not images.is_contiguous()
            and images.permute(0, 2, 3, 1).is_contiguous())


def _alloc_levels(n, h, w, device, with_level0=True):
    h1, w1 = h // 2, w // 2
    l0 = torch.empty(n, 3, h, w, device=device, dtype=torch.float32) if with_level0 else None
    l1 = torch.empty(n, 3, h1, w1, device=device, dtype=torch.float32)
    l2 = torch.empty(n, 3, h1 // 2, w1 // 2, device=device, dtype=torch.float32)
    return l0, l1, l2


# uint8 frames one sampler call can address (one 32-bit-offset buffer descriptor, csrc/warp.hip)
MAX_U8_FRAME_BYTES = (1 << 31) - 8


def build_pyramid(images_u8, materialize_level0=False, out=None):
    """uint8 [N,3,H,W] -> Pyramid: fused gamma decode (u8/255)**2.2 + two 2x2 box levels.  By
    default level 0 is NOT written as f32 (the sampler reads the uint8 frame through the LUT).
    Frames over interleaved memory (frames_are_interleaved) are used as they lie.
    out: a Pyramid made by this function over the SAME frame tensor: its levels and LUT are rewritten
    in place (fixed addresses for captured HIP graphs)."""
    require_cuda(images_u8)
    if images_u8.dtype != torch.uint8 or images_u8.ndim != 4 or images_u8.shape[1] != 3:
        raise ValueError('images must be uint8 [N,3,H,W]')
    hwc = frames_are_interleaved(images_u8) and not materialize_level0
    if not hwc:
        images_u8 = images_u8.contiguous()
    n, _, h, w = images_u8.shape
    lib = _lib.load()
    stream = current_stream_ptr(images_u8.device)
    build_u8 = lib.mtr_build_pyramid_u8_hwc if hwc else lib.mtr_build_pyramid_u8
    if materialize_level0:
        l0, l1, l2 = _alloc_levels(n, h, w, images_u8.device)
        check(lib.mtr_build_pyramid(_ptr(images_u8), n, h, w, _ptr(l0), _ptr(l1), _ptr(l2), stream),
              'mtr_build_pyramid')
        return Pyramid([l0, l1, l2])
    if out is not None:
        if out.images_u8 is None or out.images_u8.data_ptr() != images_u8.data_ptr() or \
                (out.n, out.h, out.w) != (n, h, w) or out.hwc != hwc:
            raise ValueError('build_pyramid(out=): the pyramid was not built over this frame tensor')
        check(build_u8(_ptr(images_u8), n, h, w, _ptr(out.lut), _ptr(out.levels[1]),
                       _ptr(out.levels[2]), stream), 'mtr_build_pyramid_u8')
        return out
    _, l1, l2 = _alloc_levels(n, h, w, images_u8.device, with_level0=False)
    lut = torch.empty(256, device=images_u8.device, dtype=torch.float32)
    check(build_u8(_ptr(images_u8), n, h, w, _ptr(lut), _ptr(l1), _ptr(l2), stream), 'mtr_build_pyramid_u8')
    return Pyramid([None, l1, l2], images_u8=images_u8, lut=lut, hwc=hwc)


def pyramid_from_level0(images_linear):
    """f32 linear-light [N,3,H,W] -> Pyramid (levels 1, 2 by 2x2 box filter)."""
    require_cuda(images_linear)
    l0 = images_linear.contiguous().float()
    n, _, h, w = l0.shape
    _, l1, l2 = _alloc_levels(n, h, w, l0.device, with_level0=False)
    check(_lib.load().mtr_pyramid_from_level0(_ptr(l0), n, h, w, _ptr(l1), _ptr(l2),
                                              current_stream_ptr(l0.device)),
          'mtr_pyramid_from_level0')
    return Pyramid([l0, l1, l2])


def pyramid_of_frames(images):
    """The pyramid of a call's frames [N,3,H,W] on the device.  uint8 frames: the fused LUT path (level 0 stays
    the frame).  Any other dtype -- the reference only ever says ``(images.float() / 255) ** 2.2``
    (multiperson_model.py:196), so float frames, also with non-integer values, are legal input -- by that very
    expression on the device and a materialised f32 level 0."""
    if images.dtype == torch.uint8:
        return build_pyramid(images)
    return pyramid_from_level0((images.float() / 255) ** 2.2)


def crop_geometry(boxes, intrinsics, distortion12, camspace_up, image_ids, aug_rotflipmat,
                  aug_scales, aug_gammas, res, antialias):
    """Per (aug, box): new intrinsics [A,n,3,3], R [A,n,3,3] and the warp parameter rows
    [A*n, 36] (multiperson_model.py:264-305,322-355)."""
    require_cuda(boxes, intrinsics, distortion12, camspace_up, image_ids)
    boxes = boxes.contiguous().float()
    n_box, n_aug = boxes.shape[0], aug_gammas.shape[0]
    dev = boxes.device
    new_k = torch.empty(n_aug, n_box, 3, 3, device=dev, dtype=torch.float32)
    rot = torch.empty(n_aug, n_box, 3, 3, device=dev, dtype=torch.float32)
    wp = torch.empty(n_aug * n_box, _lib.MTR_WARP_PARAM_FLOATS, device=dev, dtype=torch.float32)
    args = [intrinsics.contiguous().float(), distortion12.contiguous().float(),
            camspace_up.contiguous().float(), image_ids.contiguous().to(torch.int32),
            aug_rotflipmat.contiguous().float(), aug_scales.contiguous().float(),
            aug_gammas.contiguous().float()]
    if args[1].shape != (n_box, 12):
        raise ValueError('distortion coefficients must be zero-padded to [n_box, 12]')
    check(_lib.load().mtr_crop_geometry(
        _ptr(boxes), boxes.stride(0), *[_ptr(a) for a in args], n_box, n_aug, int(res),
        int(antialias), _ptr(new_k), _ptr(rot), _ptr(wp), current_stream_ptr(dev)),
        'mtr_crop_geometry')
    return new_k, rot, wp


def warp_crops(pyramid, warp_params, res, antialias=1, out_dtype=torch.float32,
               channels_last=False, out=None):
    """Pyramid + [n,36] warp rows -> crops [n,3,res,res] (NCHW) or NHWC memory when
    channels_last (returned as a logically-NCHW channels_last tensor)."""
    require_cuda(warp_params)
    n = warp_params.shape[0]
    dev = warp_params.device
    if antialias > 4:
        return _warp_crops_big_antialias(pyramid, warp_params, res, antialias, out_dtype, channels_last, out)
    if antialias not in (1, 2, 4):
        # (the reference warps at res*aa and only shrinks for 2, 4 and > 4: its reshape to
        #  [num_aug, n, 3, res, res] fails for 3, multiperson_model.py:307-316)
        raise ValueError(f'antialias_factor must be 1, 2, 4 or > 4 (got {antialias})')
    if out is None:
        if channels_last:
            out = torch.empty(n, res, res, 3, device=dev, dtype=out_dtype).permute(0, 3, 1, 2)
        else:
            out = torch.empty(n, 3, res, res, device=dev, dtype=out_dtype)
    l0, l1, l2 = pyramid.levels
    tail = (pyramid.n, pyramid.h, pyramid.w, _ptr(warp_params), n, int(res), int(antialias),
            dtype_code(out.dtype), _lib.MTR_NHWC if channels_last else _lib.MTR_NCHW, _ptr(out),
            current_stream_ptr(dev))
    if l0 is None:
        lib = _lib.load()
        warp_u8 = lib.mtr_warp_crops_u8_hwc if pyramid.hwc else lib.mtr_warp_crops_u8
        check(warp_u8(_ptr(pyramid.images_u8), _ptr(pyramid.lut), _ptr(l1), _ptr(l2), *tail),
              'mtr_warp_crops_u8')
    else:
        check(_lib.load().mtr_warp_crops(_ptr(l0), _ptr(l1), _ptr(l2), *tail), 'mtr_warp_crops')
    return out


def _warp_crops_big_antialias(pyramid, warp_params, res, antialias, out_dtype, channels_last, out):
    """antialias_factor > 4 (multiperson_model.py:312-315): sample at res*aa x res*aa in linear light
    (gamma exponent 1), shrink with aten's antialiased separable bilinear filter + the per-crop gamma
    (mtr_crops_shrink_antialiased).  The res*aa crops are 3 * (res*aa)^2 * 4 bytes each (50 MB at
    256 px, aa = 8): they are produced and consumed in chunks of <= 1 GiB."""
    lib = _lib.load()
    n, dev = warp_params.shape[0], warp_params.device
    if 2 * antialias + 2 > 40:
        raise ValueError(f'antialias_factor {antialias} > 19 is not supported')
    big = int(res) * int(antialias)
    if out is None:
        out = (torch.empty(n, res, res, 3, device=dev, dtype=out_dtype).permute(0, 3, 1, 2) if channels_last
               else torch.empty(n, 3, res, res, device=dev, dtype=out_dtype))
    wp_lin = warp_params.clone()
    wp_lin[:, 33] = 1.0
    per = 3 * big * big * 4
    chunk = max(1, min(n, (1 << 30) // per))
    flat_out = out.permute(0, 2, 3, 1) if channels_last else out  # the memory order, crop-major
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        tmp_big = warp_crops(pyramid, wp_lin[a:b].contiguous(), big, 1, torch.float32, False)
        ws = torch.empty(lib.mtr_crops_shrink_workspace_bytes(b - a, int(res), int(antialias)) // 4,
                         device=dev, dtype=torch.float32)
        check(lib.mtr_crops_shrink_antialiased(
            _ptr(tmp_big), _ptr(warp_params[a:b].contiguous()), b - a, int(res), int(antialias),
            dtype_code(out.dtype), _lib.MTR_NHWC if channels_last else _lib.MTR_NCHW,
            flat_out[a:b].data_ptr(), _ptr(ws), ws.numel() * 4, current_stream_ptr(dev)),
            'mtr_crops_shrink_antialiased')
    return out


def head_fused_supported(C, J, D, H, W, channels_last=False, dtype=torch.float32):
    """Shapes the fused projection+decode kernels cover; everything else goes through a library
    GEMM for the 1x1 conv followed by the HIP decode kernel (same results, logits via HBM).
    f32 features run the row-tile kernel: any map size, up to 80 depth bins.  f16 / bf16 features
    run the joint-group MFMA kernel (a joint's 1 + D rows inside a 64-row tile, maps of <= 256
    positions, C % 8 == 0) or, beyond those limits, the 16-bit row-tile kernel (C % 64 == 0, up to 80
    depth bins, any map size)."""
    if (H * W) % 4 != 0 or (channels_last and C % 4 != 0):
        return False
    if dtype == torch.float32:
        return D <= 80
    if H * W <= 256 and (1 + D) <= 64 and C % 8 == 0:
        return True
    return C % 64 == 0 and D <= 80


def head_auto_choice(C, J, D, H, W, channels_last=False, dtype=torch.float32):
    """MetrabsHeads(fused='auto'): True = the fused kernel, False = library 1x1 conv + mtr_softargmax_decode.
    A static table over (dtype, layout, C, H, W, J, D) read off the committed sweeps
    (profiles/*_fused_vs_library.txt, *_head_sweep.jsonl, r05c_f32_depth_sweep_fused_vs_library.jsonl) -- the
    batch size is deliberately NOT an input: a slice of a sharded batch, another rank and another process all
    take the path of the whole batch.  (INSIDE the fused path the library's launch plan may pick another kernel
    VARIANT by launch size -- f32 tile blocks, the 16-bit weights-in-registers kernel at >= 512 crops -- every one
    of which is bit-identical to the others on the same crop: tests/test_gpu_head.py asserts `torch.equal`
    across every dispatch choice, the default one at B = 512 against launches of 64 included.)
    The fused kernels are ahead on every shipped configuration (8x8 / 12x12 maps, 8 depth bins, 17 - 122
    joints, f32 / f16 / bf16); the library pair is kept for
      * 16-bit features on maps of more than 256 positions (the 16-bit row-tile kernel there is behind
        rocBLAS: 20x20 bf16 65 vs 39 us, 24x24 f16 37 vs 32 us),
      * f32 features on maps of >= 576 positions (24x24: 56 vs 50 us),
      * f32 features with more than 16 depth bins (round 5; the sweep at 17 joints, 8x8, B = 64 / 1024: 8 bins
        0.92 / 0.92 of the library pair's time, 16 bins 1.00 / 0.98, 24 bins 1.54 / 1.40, 32 bins 1.12 / 1.16,
        48 bins 1.14 / 1.15, 72 bins -- the metric string's shape -- 1.17 / 1.04, 80 bins 1.31 / 1.17: a
        joint's 1 + D rows no longer fit one 16-row tile and the multi-tile atoms quantise badly on 256 CUs).
        16-bit features keep the fused kernel there (72 bins, f16: 45 vs 70 us at 64 crops),
      * (round 6: the rule reads the LAYOUT) f32 channels_last features on maps of more than 64 positions: for them the
        library path is a plain GEMM (`F.linear` on the [B H W, C] matrix the features already are -- 2 - 5 x faster
        than the library's channels_last 1x1 convolution, which is what ran before) and is level with or ahead of the
        fused kernel from 12x12 maps on (64 / 256 crops of 12x12: 47 / 135 vs 52 / 158 us, 16x16 76 / 242 vs 69 / 250,
        20x20 104 / 401 vs 120 / 403; 8x8 with 8 / 16 bins stays fused: 23 / 64 vs 29 / 71, 34 / 121 vs 40 / 118;
        profiles/r06x_layout_rule*.jsonl)."""
    if not head_fused_supported(C, J, D, H, W, channels_last, dtype):
        return False
    hw = H * W
    if dtype == torch.float32:
        return (hw <= 64 if channels_last else hw < 576) and D <= 16
    return hw <= 256


def _is_channels_last(t):
    return (t.dim() == 4 and not t.is_contiguous()
            and t.is_contiguous(memory_format=torch.channels_last))


def head_pack_weights(weight2d, bias, n_points, depth, feat_dtype=torch.float32):
    """conv_final.weight [J*(1+D), C] + bias -> packed joint-major tiles for mtr_head_fused."""
    require_cuda(weight2d, bias)
    lib = _lib.load()
    weight2d = weight2d.contiguous().float()
    bias = bias.contiguous().float()
    n_out, C = weight2d.shape
    if n_out != n_points * (1 + depth):
        raise ValueError('weight rows != J*(1+depth)')
    nbytes = lib.mtr_head_packed_bytes(C, n_points, depth, dtype_code(feat_dtype))
    if nbytes == 0:
        raise ValueError(f'fused head does not support C={C}, J={n_points}, D={depth}')
    packed = torch.empty(nbytes // 4, device=weight2d.device, dtype=torch.float32)
    check(lib.mtr_head_pack_weights(_ptr(weight2d), _ptr(bias), C, n_points, depth,
                                    dtype_code(feat_dtype), _ptr(packed),
                                    current_stream_ptr(weight2d.device)), 'mtr_head_pack_weights')
    return packed


def head_fused(features, packed, C, n_points, cfg, out=None, rt_tiles=0, groups_per_workgroup=0,
               dma_staging=-1, rt_column_blocks=0, rt_k_groups=0, rt_loader=0, rt_split=0,
               workspace=None):
    """features [B,C,H,W] (f32/f16/bf16; NCHW-contiguous or torch channels_last = NHWC memory, which
    is consumed in place) -> (coords2d, coords3d_rel).  rt_tiles / groups_per_workgroup /
    dma_staging / rt_column_blocks / rt_k_groups / rt_loader / rt_split: explicit dispatch choices
    (mtr_head_options; defaults = the library's own).  workspace: None = a scratch tensor of
    mtr_head_workspace_bytes is allocated when the shape can use one (f32 maps of more than 64
    positions: column blocks over workgroups); False = none (mtr_head_fused_opts' behaviour); or a
    uint8 / float64 tensor to use."""
    require_cuda(features, packed)
    lib = _lib.load()
    nhwc = _is_channels_last(features)
    if not nhwc:
        features = features.contiguous()
    B, Cf, H, W = features.shape
    if Cf != C:
        raise ValueError(f'features have {Cf} channels, weights were packed for {C}')
    J, D = int(n_points), cfg.depth
    if out is None:
        c2d = torch.empty(B, J, 2, device=features.device, dtype=torch.float32)
        c3d = torch.empty(B, J, 3, device=features.device, dtype=torch.float32)
    else:
        c2d, c3d = out
    hp = cfg.head_params()
    layout = _lib.MTR_NHWC if nhwc else _lib.MTR_NCHW
    ws_ptr, ws_bytes = None, 0
    if workspace is not False:
        need = lib.mtr_head_workspace_bytes(dtype_code(features.dtype), layout, B, C, H, W, J, D)
        if need:
            if workspace is None:
                workspace = torch.empty((need + 7) // 8, device=features.device, dtype=torch.float64)
            require_cuda(workspace)
            ws_ptr, ws_bytes = _ptr(workspace), workspace.numel() * workspace.element_size()
    opts = _lib.head_options(rt_tiles, groups_per_workgroup, dma_staging, rt_column_blocks, rt_k_groups,
                             rt_loader, rt_split)
    check(lib.mtr_head_fused_ws(
        _ptr(features), dtype_code(features.dtype), layout, B, C, H, W, _ptr(packed), J, D,
        ctypes.byref(hp), ctypes.byref(opts), ws_ptr, ws_bytes, _ptr(c2d), _ptr(c3d),
        current_stream_ptr(features.device)), 'mtr_head_fused_ws')
    return c2d, c3d


def head_plan(B, C, H, W, n_points, depth, dtype=torch.float32, channels_last=False, have_workspace=True,
              **options):
    """Which kernel mtr_head_fused_ws takes for a launch (host-only; mtr_head_plan) -> dict(kernel=name,
    tiles_per_workgroup, column_blocks, split_column_blocks, workgroups, model_us) or None when the shape has
    no fused kernel.  options: as head_fused."""
    lib = _lib.load()
    opts = _lib.head_options(**{k: options[k] for k in ('rt_tiles', 'groups_per_workgroup', 'dma_staging',
                                                        'rt_column_blocks', 'rt_k_groups', 'rt_loader', 'rt_split')
                                if k in options})
    info = _lib.HeadPlanInfo()
    rc = lib.mtr_head_plan(dtype_code(dtype), _lib.MTR_NHWC if channels_last else _lib.MTR_NCHW, int(B), int(C),
                           int(H), int(W), int(n_points), int(depth), ctypes.byref(opts), int(bool(have_workspace)),
                           ctypes.byref(info))
    if rc != 0:
        return None
    return dict(kernel=_lib.HEAD_KERNEL_NAMES.get(info.kernel, str(info.kernel)),
                tiles_per_workgroup=info.tiles_per_workgroup, column_blocks=info.column_blocks,
                split_column_blocks=info.split_column_blocks, workgroups=int(info.workgroups),
                model_us=float(info.model_us))


def postprocess_poses(poses_crop, rot, should_flip, mirror_mapping, intrinsics, distortion12,
                      inv_extrinsics, joint_transform=None, skeleton=None, average_aug=True):
    """K7: crop-model output [A*n, J, 3] (or [A,n,J,3]) + R [A,n,3,3] -> (poses3d, poses2d) in the
    original camera / world frame: [n, S, 3|2] if average_aug else [n, A, S, 3|2]
    (multiperson_model.py:143-178,244-259)."""
    require_cuda(poses_crop, rot, intrinsics)
    dev = poses_crop.device
    A, n = rot.shape[0], rot.shape[1]
    poses_crop = poses_crop.contiguous().float().reshape(A, n, -1, 3)
    J = poses_crop.shape[2]
    # (.to is a no-op for tensors that already have the kernel's dtype on the device; callers on a
    # hot path pass cached uint8 / int32 device tensors)
    flip = should_flip.to(dev, torch.uint8).contiguous()
    mirror = mirror_mapping.to(dev, torch.int32).contiguous()
    jtm = joint_transform.to(dev, torch.float32).contiguous() if joint_transform is not None else None
    skel = skeleton.to(dev, torch.int32).contiguous() if skeleton is not None else None
    Jt = jtm.shape[1] if jtm is not None else J
    S = skel.numel() if skel is not None else Jt
    shape = (n, S) if average_aug else (n, A, S)
    p3 = torch.empty(*shape, 3, device=dev, dtype=torch.float32)
    p2 = torch.empty(*shape, 2, device=dev, dtype=torch.float32)
    check(_lib.load().mtr_postprocess_poses(
        _ptr(poses_crop), _ptr(rot.contiguous().float()), _ptr(flip), _ptr(mirror), _ptr(jtm), Jt,
        _ptr(skel), S, _ptr(intrinsics.contiguous().float()), _ptr(distortion12.contiguous().float()),
        _ptr(inv_extrinsics.contiguous().float()), A, n, J, int(bool(average_aug)), _ptr(p3),
        _ptr(p2), current_stream_ptr(dev)), 'mtr_postprocess_poses')
    return p3, p2


def linear_combine_points(points, weights, out=None):
    """points [B, J_in, 3], weights [J_in, J_out] -> [B, J_out, 3]: einsum 'bjc,jJ->bJc'
    (tfu3d.linear_combine_points, metrabs_tf/tfu3d.py:48-49; Metrabs.latent_points_to_joints,
    metrabs_tf/models/metrabs.py:80-81)."""
    require_cuda(points, weights)
    points = points.contiguous().float()
    weights = weights.contiguous().float()
    B, j_in = points.shape[:2]
    if points.shape[2] != 3 or weights.dim() != 2 or weights.shape[0] != j_in:
        raise ValueError(f'points {tuple(points.shape)} and weights {tuple(weights.shape)} do not match')
    j_out = weights.shape[1]
    if out is None:
        out = torch.empty(B, j_out, 3, device=points.device, dtype=torch.float32)
    if B == 0:
        return out
    check(_lib.load().mtr_linear_combine_points(_ptr(points), _ptr(weights), B, j_in, j_out, _ptr(out),
                                                current_stream_ptr(points.device)),
          'mtr_linear_combine_points')
    return out


# ------------------------------------------------------------------------------------------------
# K9: detector pre-processing (person_detector.py:14-54 minus the network)

def detector_geometry(h, w, input_size=416):
    """person_detector.py:15-20,26-29 -> _lib.DetectorGeom (host arithmetic, no GPU work)."""
    g = _lib.DetectorGeom()
    check(_lib.load().mtr_detector_geometry(int(h), int(w), int(input_size), ctypes.byref(g)),
          'mtr_detector_geometry')
    return g


DETECTOR_KERNELS = {'auto': 0, 'tile': 1, 'stream': 2}  # MTR_DETECTOR_KERNEL_*


def detector_preprocess(images_u8, geom=None, input_size=416, out=None, kernel='auto'):
    """images_u8 [N,3,H,W] uint8 (cuda) -> ([N,3,out_h,out_w] f32 as fed to the detector network,
    geometry).  person_detector.py:21-33.  `kernel`: 'auto' | 'tile' | 'stream' (identical bits; tests
    and timing name one)."""
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[1] != 3:
        raise ValueError('images must be uint8 [N,3,H,W]')
    require_cuda(images_u8)
    images_u8 = images_u8.contiguous()
    N, _, H, W = images_u8.shape
    g = geom if geom is not None else detector_geometry(H, W, input_size)
    if out is None:
        out = torch.empty(N, 3, g.out_h, g.out_w, device=images_u8.device, dtype=torch.float32)
    check(_lib.load().mtr_detector_preprocess_kernel(_ptr(images_u8), N, H, W, ctypes.byref(g),
                                                     DETECTOR_KERNELS[kernel], _ptr(out),
                                                     current_stream_ptr(images_u8.device)),
          'mtr_detector_preprocess_kernel')
    return out, g


def detector_scale_boxes(xyxy_conf, geom):
    """[n,5] (x1,y1,x2,y2,conf) in the padded network frame -> [n,5] (x,y,w,h,conf) in the image
    frame.  person_detector.py:47-54."""
    require_cuda(xyxy_conf)
    b = xyxy_conf.float().contiguous()
    out = torch.empty_like(b)
    check(_lib.load().mtr_detector_scale_boxes(_ptr(b), b.shape[0], ctypes.byref(geom), _ptr(out),
                                               current_stream_ptr(b.device)),
          'mtr_detector_scale_boxes')
    return out


# ------------------------------------------------------------------------------------------------
# K8: plausibility filter + pose NMS (plausibility_check.py / TF _filter_poses)

def filter_poses(poses3d, poses2d, boxes, n_per_image, edges, mean_bones, n_joints=None,
                 unbiased=False, order='index', max_output=150):
    """poses3d [P,A,J,3] (camera space, all model joints), poses2d [P,A,J,2], boxes [P,5], poses of
    image i contiguous with n_per_image[i] rows.  edges [n_bones,2] / mean_bones [n_bones] may be
    None (no bone-length test).  -> (keep_idx [P] i32: per image its kept rows then -1,
    keep_count [n_images] i32, valid [P] bool) -- all on the GPU, no host sync."""
    require_cuda(poses3d, poses2d, boxes)
    P, A, J, _ = poses3d.shape
    dev = poses3d.device
    n_images = len(n_per_image)
    counts = [int(x) for x in n_per_image]
    if sum(counts) != P:
        raise ValueError('n_per_image does not add up to the number of poses')
    row_start = torch.tensor([0] + list(torch.tensor(counts).cumsum(0).tolist()) if counts else [0],
                             dtype=torch.int32, device=dev)
    sq = [c * c for c in counts]
    sim_off = torch.tensor([sum(sq[:i]) for i in range(n_images)], dtype=torch.int32, device=dev)
    max_n = max(counts) if counts else 0
    lib = _lib.load()
    ws_bytes = lib.mtr_filter_poses_workspace_bytes(P, J, max_n)
    ws = torch.empty(ws_bytes // 8 + 1, dtype=torch.float64, device=dev)
    valid = torch.zeros(P, dtype=torch.uint8, device=dev)
    keep_idx = torch.full((P,), -1, dtype=torch.int32, device=dev)
    keep_count = torch.zeros(n_images, dtype=torch.int32, device=dev)
    if edges is not None and len(edges):
        e = torch.as_tensor(edges, dtype=torch.int32, device=dev).contiguous()
        mb = torch.as_tensor(mean_bones, dtype=torch.float32, device=dev).contiguous()
        n_bones = e.shape[0]
    else:
        e = mb = None
        n_bones = 0
    fp = _lib.FilterParams(0.1, 3.0, 300.0, 200.0, 0.5, 300.0, 0.4, int(max_output),
                           1 if unbiased else 0, 1 if order == 'score' else 0)
    p3 = poses3d.float().contiguous()
    p2 = poses2d.float().contiguous()
    bx = boxes.float().contiguous()
    check(lib.mtr_filter_poses(_ptr(p3), _ptr(p2), _ptr(bx), _ptr(row_start), _ptr(sim_off), n_images,
                               P, A, J, _ptr(e) if e is not None else None,
                               _ptr(mb) if mb is not None else None, n_bones,
                               int(n_joints or J), ctypes.byref(fp), _ptr(ws), ws.numel() * 8, max_n,
                               _ptr(valid), _ptr(keep_idx), _ptr(keep_count),
                               current_stream_ptr(dev)), 'mtr_filter_poses')
    return keep_idx, keep_count, valid.bool()


# ------------------------------------------------------------------------------------------------
# K10: bias + activation epilogue for the backbone's inference copy (backbones.fold_batchnorm)

ACT_CODES = {None: 0, 'none': 0, 'relu': 1, 'silu': 2, 'hardswish': 3}


def bias_act_(y, bias, act, residual=None):
    """In place: y[b, c, ...] = act(y[b, c, ...] + bias[c]) (+ residual) on an NCHW-contiguous
    activation (f32 / f16 / bf16), one HBM pass instead of PyTorch-ROCm's bias-add, activation and
    skip-connection kernels."""
    require_cuda(y, bias)
    if not y.is_contiguous():
        raise ValueError('bias_act_ needs an NCHW-contiguous tensor')
    if residual is not None:
        if residual.shape != y.shape or residual.dtype != y.dtype or not residual.is_contiguous():
            raise ValueError('residual must match y in shape, dtype and layout')
    B, C = y.shape[0], y.shape[1]
    hw = y.numel() // max(B * C, 1)
    check(_lib.load().mtr_bias_act_nchw(_ptr(y), dtype_code(y.dtype), _ptr(bias.contiguous().float()),
                                        None if residual is None else _ptr(residual), ACT_CODES[act],
                                        B, C, hw, current_stream_ptr(y.device)),
          'mtr_bias_act_nchw')
    return y


def bias_act_rowmean_(y, bias, act):
    """bias_act_ + the mean over H*W of every (b, c) row of the result, [B, C] f32, from the same pass
    (what a squeeze-excite block behind the convolution starts with)."""
    require_cuda(y, bias)
    if not y.is_contiguous():
        raise ValueError('bias_act_rowmean_ needs an NCHW-contiguous tensor')
    B, C = y.shape[0], y.shape[1]
    hw = y.numel() // max(B * C, 1)
    mean = torch.empty(B, C, device=y.device, dtype=torch.float32)
    check(_lib.load().mtr_bias_act_rowmean_nchw(_ptr(y), dtype_code(y.dtype), _ptr(bias.contiguous().float()),
                                                ACT_CODES[act], B, C, hw, _ptr(mean),
                                                current_stream_ptr(y.device)),
          'mtr_bias_act_rowmean_nchw')
    return y, mean


def depthwise3x3_bias_act(x, weight, bias, act, stride, pad, want_mean=False):
    """K11: y = act(depthwise_conv3x3(x, weight) + bias) in one pass (+ the [B, C] f32 mean of y over
    H*W if want_mean).  x NCHW-contiguous f32 / f16 / bf16, weight [C, 1, 3, 3] or [C, 3, 3].
    pad: one int (every side) or (left, right, top, bottom) like torch.nn.ZeroPad2d -- the explicit
    padding in front of the reference's stride-2 layers, folded into the kernel."""
    require_cuda(x, weight, bias)
    if not x.is_contiguous():
        raise ValueError('depthwise3x3_bias_act needs an NCHW-contiguous tensor')
    B, C, H, W = x.shape
    pl, pr, pt, pb = (pad,) * 4 if isinstance(pad, int) else tuple(int(p) for p in pad)
    OH, OW = (H + pt + pb - 3) // stride + 1, (W + pl + pr - 3) // stride + 1
    y = torch.empty(B, C, OH, OW, device=x.device, dtype=x.dtype)
    mean = torch.empty(B, C, device=x.device, dtype=torch.float32) if want_mean else None
    check(_lib.load().mtr_depthwise3x3_bias_act_padded(
        _ptr(x), dtype_code(x.dtype), _ptr(weight.contiguous().float()), _ptr(bias.contiguous().float()),
        ACT_CODES[act], B, C, H, W, int(stride), pt, pl, pb, pr, _ptr(y), None if mean is None else _ptr(mean),
        current_stream_ptr(x.device)), 'mtr_depthwise3x3_bias_act_padded')
    return (y, mean) if want_mean else y
