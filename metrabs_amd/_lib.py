"""ctypes binding of libmetrabs_hip.so (the C-ABI declared in include/metrabs_hip.h).

The product path has NO CPU fallback: if the library is missing, or a kernel entry point returns an
error, a RuntimeError is raised.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint8, c_void_p

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libmetrabs_hip.so')

MTR_F32, MTR_F16, MTR_BF16 = 0, 1, 2
MTR_NCHW, MTR_NHWC = 0, 1
MTR_WARP_PARAM_FLOATS = 36


class HeadParams(ctypes.Structure):
    """mtr_head_params (include/metrabs_hip.h)."""
    _fields_ = [('proc_side', c_int32), ('stride_test', c_int32), ('centered_stride', c_int32),
                ('legacy_centered_stride_bug', c_int32), ('box_size_mm', c_float)]


class HeadOptions(ctypes.Structure):
    """mtr_head_options (include/metrabs_hip.h): explicit dispatch choices of mtr_head_fused_opts."""
    _fields_ = [('struct_size', ctypes.c_uint32),
                ('rt_tiles_per_workgroup', c_int32), ('groups_per_workgroup', c_int32),
                ('dma_staging', c_int32), ('rt_column_blocks', c_int32), ('rt_k_groups', c_int32),
                ('rt_loader', c_int32), ('rt_split_column_blocks', c_int32)]


def head_options(rt_tiles=0, groups_per_workgroup=0, dma_staging=-1, rt_column_blocks=0, rt_k_groups=0,
                 rt_loader=0, rt_split=0):
    """A versioned mtr_head_options (struct_size = this binding's sizeof)."""
    return HeadOptions(ctypes.sizeof(HeadOptions), int(rt_tiles), int(groups_per_workgroup), int(dma_staging),
                       int(rt_column_blocks), int(rt_k_groups), int(rt_loader), int(rt_split))


class HeadPlanInfo(ctypes.Structure):
    """mtr_head_plan_info (include/metrabs_hip.h): which kernel mtr_head_fused_ws takes for a launch."""
    _fields_ = [('kernel', c_int32), ('tiles_per_workgroup', c_int32), ('column_blocks', c_int32),
                ('split_column_blocks', c_int32), ('workgroups', ctypes.c_int64), ('model_us', ctypes.c_double)]


HEAD_KERNEL_NAMES = {1: 'head_rt_kernel', 2: 'head_rt_ld_kernel', 3: 'head_rt_ks_kernel', 4: 'head_rt_np_kernel',
                     10: 'head_fused16_kernel', 11: 'head_fused16dma_kernel', 12: 'head_rt16_kernel',
                     13: 'head_fused16dma_kernel (loader wave)', 14: 'head_fused16dma_kernel (early copies)',
                     15: 'head_fused16areg_kernel (weights in registers)',
                     16: 'head_fused16res_kernel (weights resident, persistent workgroups)',
                     17: 'head_fused16pp_kernel (two alternating halves)',
                     18: 'head_fused16dma_kernel (early copies, tight stage)'}


class ReconParams(ctypes.Structure):
    """mtr_recon_params (include/metrabs_hip.h)."""
    _fields_ = [('proc_side', c_int32), ('stride_train', c_int32), ('centered_stride', c_int32),
                ('weak_perspective', c_int32), ('mix_enabled', c_int32),
                ('mix_3d_inside_fov', c_float), ('l2_reg', c_float), ('weight_eps', c_float),
                ('fov_border_factor', c_float)]


class DetectorGeom(ctypes.Structure):
    """mtr_detector_geom (include/metrabs_hip.h)."""
    _fields_ = [('target_h', c_int32), ('target_w', c_int32), ('antialias', c_int32),
                ('pad_top', c_int32), ('pad_left', c_int32), ('out_h', c_int32), ('out_w', c_int32),
                ('x_factor', c_float), ('y_factor', c_float)]


class FilterParams(ctypes.Structure):
    """mtr_filter_params (include/metrabs_hip.h); defaults = the reference's constants."""
    _fields_ = [('rel_small', c_float), ('rel_big', c_float), ('abs_diff_mm', c_float),
                ('stdev_mm', c_float), ('box_fraction', c_float), ('sim_scale_mm', c_float),
                ('sim_threshold', c_float), ('max_output', c_int32), ('var_correction', c_int32),
                ('order_by_score', c_int32)]


# name -> (restype, argtypes); must list EVERY symbol the header declares
# (tests/test_capi_symbols.py cross-checks this table against include/metrabs_hip.h).
SIGNATURES = {
    'mtr_version': (c_int, []),
    'mtr_strerror': (c_char_p, [c_int]),
    'mtr_softargmax_decode': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      POINTER(HeadParams), c_void_p, c_void_p, c_void_p]),
    'mtr_softargmax_decode_opts': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                           POINTER(HeadParams), c_int, c_void_p, c_void_p, c_void_p]),
    'mtr_head_row_plan': (c_int, [c_int, c_int, POINTER(c_int32), POINTER(c_int32), c_void_p, c_int]),
    'mtr_head_packed_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'mtr_head_pack_weights': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                      c_void_p]),
    'mtr_head_fused': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                               c_int, POINTER(HeadParams), c_void_p, c_void_p, c_void_p]),
    'mtr_head_fused_opts': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                    c_int, POINTER(HeadParams), POINTER(HeadOptions), c_void_p, c_void_p,
                                    c_void_p]),
    'mtr_head_plan': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(HeadOptions),
                              c_int, POINTER(HeadPlanInfo)]),
    'mtr_head_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'mtr_head_fused_ws': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                  c_int, POINTER(HeadParams), POINTER(HeadOptions), c_void_p, c_size_t,
                                  c_void_p, c_void_p, c_void_p]),
    'mtr_crops_shrink_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'mtr_crops_shrink_antialiased': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                             c_void_p, c_size_t, c_void_p]),
    'mtr_reconstruct_workspace_bytes': (c_size_t, [c_int, c_int]),
    'mtr_reconstruct_absolute': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int,
                                         POINTER(ReconParams), c_void_p, c_void_p, c_size_t,
                                         c_void_p]),
    'mtr_reconstruct_moments': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                        c_void_p, c_size_t, c_void_p]),
    'mtr_reconstruct_solve': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      POINTER(ReconParams), c_void_p, c_void_p, c_void_p]),
    'mtr_build_pyramid': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    'mtr_build_pyramid_u8': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    'mtr_warp_crops_u8': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                  c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'mtr_build_pyramid_u8_hwc': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p]),
    'mtr_warp_crops_u8_hwc': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                      c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'mtr_pyramid_from_level0': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'mtr_crop_geometry': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    'mtr_postprocess_poses': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                      c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_void_p]),
    'mtr_linear_combine_points': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'mtr_warp_crops': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                               c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'mtr_detector_geometry': (c_int, [c_int, c_int, c_int, POINTER(DetectorGeom)]),
    'mtr_detector_preprocess': (c_int, [c_void_p, c_int, c_int, c_int, POINTER(DetectorGeom), c_void_p,
                                        c_void_p]),
    'mtr_detector_preprocess_kernel': (c_int, [c_void_p, c_int, c_int, c_int, POINTER(DetectorGeom), c_int,
                                               c_void_p, c_void_p]),
    'mtr_detector_scale_boxes': (c_int, [c_void_p, c_int, POINTER(DetectorGeom), c_void_p, c_void_p]),
    'mtr_filter_poses_workspace_bytes': (ctypes.c_size_t, [c_int, c_int, c_int]),
    'mtr_filter_poses': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                 c_int, c_void_p, c_void_p, c_int, c_int, POINTER(FilterParams), c_void_p,
                                 ctypes.c_size_t, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mtr_depthwise3x3_bias_act': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, ctypes.c_longlong,
                                          c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'mtr_depthwise3x3_bias_act_padded': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, ctypes.c_longlong,
                                                 c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                 c_void_p, c_void_p, c_void_p]),
    'mtr_bias_act_rowmean_nchw': (c_int, [c_void_p, c_int, c_void_p, c_int, ctypes.c_longlong, c_int, c_int,
                                          c_void_p, c_void_p]),
    'mtr_bias_act_nchw': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, ctypes.c_longlong, c_int,
                                  c_int, c_void_p]),
}

_lib = None


def load(path=None):
    """Loads the library and declares every prototype.  Raises if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    # torch bundles its own libamdhip64.so.7; importing it FIRST makes the loader resolve our
    # NEEDED libamdhip64.so.7 to that already-loaded runtime, so kernels launched here and torch's
    # streams/allocations live in ONE HIP runtime.  Loading this library first would pull in
    # /opt/rocm's copy instead and launches would fail with hipErrorNoDevice.
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise RuntimeError(
            f'{path} not found: the HIP extension is not built. Run `python -m metrabs_amd.build` '
            f'(needs hipcc). metrabs_amd has no CPU fallback.')
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().mtr_strerror(code).decode()
        raise RuntimeError(f'{what} failed: {msg} (code {code})')


def dtype_code(torch_dtype):
    import torch
    try:
        return {torch.float32: MTR_F32, torch.float16: MTR_F16, torch.bfloat16: MTR_BF16}[torch_dtype]
    except KeyError:
        raise TypeError(f'unsupported dtype {torch_dtype}') from None


def current_stream_ptr(device=None):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                'metrabs_amd kernels run on the GPU only (tensor on %s); there is no CPU fallback'
                % t.device)
