/*
 * metrabs_hip.h -- C-ABI of the MI355X-native MeTRAbs per-crop hot path (libmetrabs_hip.so).
 *
 * The reference (isarandi/metrabs) is pure Python and has no FFI boundary; the path is reached by
 * ordinary method calls.  Each entry point below replaces the reference
 * call(s) cited next to it; the Python drop-ins in metrabs_amd/ (same class / function names as the
 * reference) are the only callers, via ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless it says "host";
 *   - nothing allocates, synchronises or touches global state: every call only enqueues work on
 *     `stream` (a hipStream_t), so all of them are hipGraph-capturable and re-entrant;
 *   - return value: 0 = ok, <0 = argument error detected on the host before any launch
 *     (MTR_E_*), >0 = hipError_t of a failed launch.  Nothing throws or exits across the ABI;
 *   - tensors are dense, row-major in the index order written in the comment.
 */
#ifndef METRABS_HIP_H_
#define METRABS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTR_VERSION 100 /* 0.1.0 */

typedef void* mtr_stream_t; /* hipStream_t */

enum mtr_dtype { MTR_F32 = 0, MTR_F16 = 1, MTR_BF16 = 2 };

enum mtr_layout {
  MTR_NCHW = 0, /* metrabs_pytorch: [B, C, H, W]            (models/metrabs.py:79 'b (d j) h w') */
  MTR_NHWC = 1  /* metrabs_tf / torch channels_last: [B,H,W,C] (tf models/metrabs.py:100-101)      */
};

enum mtr_error {
  MTR_OK = 0,
  MTR_E_NULL = -1,      /* a required pointer is NULL                      */
  MTR_E_SHAPE = -2,     /* a dimension is <= 0 or out of the supported range */
  MTR_E_DTYPE = -3,     /* unsupported dtype / layout combination           */
  MTR_E_PARAM = -4,     /* inconsistent params struct                       */
  MTR_E_WORKSPACE = -5, /* workspace too small / misaligned                 */
  MTR_E_ALIGN = -6      /* a pointer violates the documented alignment      */
};

/* The flags the reference reads from its global config at call time
 * (metrabs_pytorch/config/config.yaml:1-22, config_s_256.yaml:5-9; read in models/util.py:7,24,30). */
typedef struct mtr_head_params {
  int32_t proc_side;                  /* FLAGS.proc_side (256 / 384)                          */
  int32_t stride_test;                /* FLAGS.stride_test: inference stride of heatmap_to_image */
  int32_t centered_stride;            /* FLAGS.centered_stride                                 */
  int32_t legacy_centered_stride_bug; /* FLAGS.legacy_centered_stride_bug                      */
  float box_size_mm;                  /* FLAGS.box_size_mm (2200)                              */
} mtr_head_params;

/* Flags / constants of ptu3d.reconstruct_absolute and friends (ptu3d.py:9-33,56-121). */
typedef struct mtr_recon_params {
  int32_t proc_side;         /* FLAGS.proc_side                                   */
  int32_t stride_train;      /* FLAGS.stride_train (is_within_fov, ptu3d.py:115)  */
  int32_t centered_stride;   /* FLAGS.centered_stride (ptu3d.py:116)              */
  int32_t weak_perspective;  /* FLAGS.weak_perspective (ptu3d.py:14-18)           */
  int32_t mix_enabled;       /* 0 <=> mix_3d_inside_fov is None (ptu3d.py:28)     */
  float mix_3d_inside_fov;   /* FLAGS.mix_3d_inside_fov (0.5)                     */
  float l2_reg;              /* 1e-2   (ridge rows sqrt(1e-2), ptu3d.py:96-97)    */
  float weight_eps;          /* 1e-4   (weights = mask + 1e-4, ptu3d.py:94)       */
  float fov_border_factor;   /* 0.75   (is_within_fov default, ptu3d.py:113)      */
} mtr_recon_params;

int mtr_version(void);
const char* mtr_strerror(int code);

/* ------------------------------------------------------------------------------------------------
 * K2-K4: volumetric soft-argmax decode.  Replaces, for logits already in memory,
 *   MetrabsHeads.forward after the 1x1 conv   metrabs_pytorch/models/metrabs.py:78-85
 *   ptu.softmax / ptu.decode_heatmap          metrabs_pytorch/ptu.py:47-75
 *   heatmap_to_image / heatmap_to_metric      metrabs_pytorch/models/util.py:6-33
 *
 * logits   [B, J*(1+D), H, W] (MTR_NCHW) or [B, H, W, J*(1+D)] (MTR_NHWC), dtype f32/f16/bf16.
 *          channel n < J: 2D heatmap of joint n; channel J + d*J + j: depth slice d of joint j.
 * coords2d     [B, J, 2] f32, pixels (x, y) of the crop
 * coords3d_rel [B, J, 3] f32, millimetres (x, y, z) relative to the unknown reference point
 */
int mtr_softargmax_decode(const void* logits, int dtype, int layout, int B, int J, int D, int H,
                          int W, const mtr_head_params* p, float* coords2d, float* coords3d_rel,
                          mtr_stream_t stream);
/* The same launch with the NHWC kernel choice made explicit (A/B measurements, the bit-identity tests):
 * nhwc_staging 0 = the library's rule, 1 = never, 2 = whenever the shape allows, 3 = as 2 with two crops per workgroup
 * where that fills the waves better (measured, never the rule's choice) -- the kernel that streams a crop's logits
 * through a ring of LDS slots filled by 16-byte-per-lane global_load_lds and walks them there (by the rule: launches of
 * >= 256 crops of <= 1,024 channels).  Same operations in the same order: the same bits.  Ignored for MTR_NCHW. */
int mtr_softargmax_decode_opts(const void* logits, int dtype, int layout, int B, int J, int D, int H,
                               int W, const mtr_head_params* p, int nhwc_staging, float* coords2d,
                               float* coords3d_rel, mtr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K1+K2-K4 fused: 1x1 heatmap-projection GEMM (MFMA) with the decode as its epilogue; the logits
 * never reach HBM.  Replaces MetrabsHeads.forward as a whole (models/metrabs.py:75-85; the conv is
 * torch.nn.LazyConv2d(kernel_size=1), models/metrabs.py:73).
 *
 * mtr_head_pack_weights re-orders conv_final.weight [J*(1+D), C] (+bias) ONCE into the tiled
 * layout the kernel streams; `packed` must hold mtr_head_packed_bytes(...) bytes
 * (0 = no fused kernel for this (C, J, D, dtype): run the 1x1 conv as a GEMM and
 * mtr_softargmax_decode on its logits).  The blob is specific to the feat_dtype it was packed for:
 * pass the same feat_dtype to mtr_head_packed_bytes, mtr_head_pack_weights and mtr_head_fused.
 * features: [B, C, H, W] (MTR_NCHW) or [B, H, W, C] (MTR_NHWC, C % 4 == 0); H*W a multiple of 4.
 *
 * f32 features (the reference's CPU arithmetic) -- row-tile kernel, any map size, D <= 80:
 *   v_mfma_f32_16x16x4_f32 on f32 weights, chains of 32 channels summed in f32 over 8 stages and
 *   carried into f64 accumulators; 16-row tiles of whole softmax units (mtr_head_row_plan), a
 *   workgroup = (crop, block of <= 5 tiles), tiles staged by global_load_lds -- on launches of at most
 *   one workgroup per CU by a dedicated loader wave (mtr_head_options.rt_loader); with a workspace
 *   (mtr_head_fused_ws) the 64-position column blocks of a larger map may go to different workgroups
 *   (rt_split_column_blocks).  Every dispatch choice gives the same bits.
 * f16 / bf16 features (the reference's autocast GPU arithmetic) -- joint-group kernels, C % 8 == 0,
 *   1 + D <= 64, H*W <= 256: v_mfma_f32_32x32x16_{f16,bf16} on the features and on the weights
 *   ROUNDED TO THE FEATURE DTYPE (what autocast does to conv_final), f32 accumulation, f32 logits.
 *   (C % 64 == 0, and for NCHW also H*W % 8 == 0 and H*W >= 64: tiles staged by global_load_lds
 *   instead of through registers -- same arithmetic, same results.)
 *   Beyond those limits (72 depth bins; maps of more than 256 positions), C % 64 == 0, D <= 80: the
 *   row-tile core on v_mfma_f32_16x16x32_{f16,bf16} (head_rt16_kernel), any map size.  It consumes
 *   NHWC features; NCHW features are transposed once into the workspace of mtr_head_fused_ws
 *   (without one: MTR_E_WORKSPACE).
 *
 * mtr_head_fused_opts: the same launch with explicit dispatch choices (A/B measurements, tests of
 * every kernel variant); options == NULL or all-zero fields = the library's own choice.  There are
 * no environment switches behind these entry points, and no state that changes a result.  What the
 * library does keep per process is MEMOISATION of pure functions of its arguments: the launch plans of
 * the f32 head (a mutex-guarded map keyed by the current device's CU count, the shape -- with the batch
 * size rounded up to a multiple of 8, all the plan reads of it -- and the options), the per-kernel "dynamic LDS allowed" once-flags, and
 * the detector pre-processing's per-shape kernel choice; all of it is rebuilt identically on a miss.
 *
 * mtr_head_fused / mtr_head_fused_opts have no workspace to offer: a shape that needs one (16-bit NCHW
 * features on the row-tile core) is reported as MTR_E_SHAPE -- "take the library-GEMM path" -- exactly
 * like any other shape without a fused kernel; only mtr_head_fused_ws returns MTR_E_WORKSPACE.
 *
 * mtr_head_options is versioned by its first member: set struct_size = sizeof(mtr_head_options) of the
 * header you compiled against.  The library reads the fields inside struct_size and takes its own choice
 * for the ones beyond it (a caller built against an older, shorter struct keeps working); a struct_size
 * that is not a multiple of 4, below 8 or above 256 is MTR_E_PARAM (a round-3 caller, whose struct began
 * with rt_tiles_per_workgroup = 0..5, is rejected instead of being misread).
 */
typedef struct mtr_head_options {
  uint32_t struct_size;            /* sizeof(mtr_head_options) as the CALLER knows it (see above)            */
  int32_t rt_tiles_per_workgroup;  /* f32: row tiles per workgroup for one-tile atoms, 1..5; 0 = by launch size */
  int32_t groups_per_workgroup;    /* 16-bit: joint groups per workgroup, 1..3 (dma_staging 4: 2..4 = waves
                                      per workgroup); 0 = by launch size                                    */
  int32_t dma_staging;             /* 16-bit: 1 = global_load_lds where possible, 0 = through registers,
                                      2 = global_load_lds issued by a fifth, LOADER wave (320 threads; same
                                      bits; measured 10 - 25 % SLOWER than 1 on every shape of
                                      profiles/r04c_head16_ab.jsonl -- kept for A/B runs, never the library's
                                      choice), 3 = global_load_lds with the copies of stage s + 1 issued first in
                                      stage s and the fragments read per 16-channel step (round 4; same bits),
                                      4 = weights in registers (round 5, head_areg.hip: a wave per joint group
                                      against all column tiles, weights loaded per lane from the fragment-major
                                      section of the blob, only the features through LDS; 3 - 5 column tiles,
                                      C % 64 == 0; same bits; elsewhere: as -1),
                                      5 = weights RESIDENT in registers, persistent workgroups (round 5,
                                      head_res.hip: a workgroup keeps a pair of joint groups' weights for the
                                      whole launch and streams its share of the crops through a ring of feature
                                      stages; C = 1280, 3 - 5 column tiles; same bits; 20 - 40 % SLOWER than the
                                      library's choice on every shape of profiles/r05y_* -- its decode epilogue
                                      has no second workgroup's K loop to hide behind; kept for A/B runs;
                                      elsewhere: as -1),
                                      6 = eight waves in two alternating halves (round 6, head_pp.hip: four joint
                                      groups per workgroup; one half multiplies a 64-channel stage while the other
                                      issues the next stage's copies; C % 64 == 0, 3 - 5 column tiles; same bits;
                                      elsewhere: as -1),
                                      7 = early copies on a tight feature stage (round 6: H*W positions instead of
                                      whole 32-position tiles, exactly two stages of LDS; groups_per_workgroup 1
                                      (default) or 2; H*W % 8 == 0; same bits; elsewhere: as -1),
                                      -1 = library's choice (NB: a zeroed struct selects registers) */
  int32_t rt_column_blocks;        /* f32, maps of > 64 positions: 64-position column blocks per workgroup
                                      tile, 2..4 (one K loop for all of them); 1 = one K loop per column
                                      block; 0 = by launch size                                         */
  int32_t rt_k_groups;             /* f32, blocks of <= 3 one-tile atoms, C % 64 == 0: 2 = two K groups per
                                      workgroup (512 threads: the odd 32-channel stages run on waves
                                      4..7 and their chains are handed to waves 0..3, which sum in the
                                      one-group order: same bits), 1 = one; 0 = two for blocks of 2-3
                                      tiles (the small-launch configuration)                          */
  int32_t rt_loader;               /* f32, C % 32 == 0: 2 = four MFMA waves + a LOADER wave per workgroup (320
                                      threads; the fifth wave issues every global_load_lds and waits for it,
                                      the MFMA waves only meet it at the stage barrier: same bits), 1 = never;
                                      0 = for launches of at most one workgroup per CU                 */
  int32_t rt_split_column_blocks;  /* f32, maps of > 64 positions, mtr_head_fused_ws with a workspace: 2 = deal
                                      the 64-position column blocks to different workgroups (a second, tiny
                                      launch merges their softmax statistics in block order: same bits as one
                                      workgroup walking them), 1 = never; 0 = on small launches        */
} mtr_head_options;

/* host-only, no GPU work: the row order of the f32 row-tile kernel.  conv_final's J*(1+D) channels
 * are re-ordered into n_tiles 16-row MFMA tiles so that the rows of one softmax (a joint's 2D row;
 * a joint's D depth slices) never straddle a workgroup's block of tiles; atoms of tiles_per_atom
 * consecutive tiles are the unit a block is made of.  row_channel (may be NULL; `capacity` ints,
 * >= n_tiles*16) receives for every packed row the conv_final channel it holds, -1 for padding. */
int mtr_head_row_plan(int J, int D, int32_t* n_tiles, int32_t* tiles_per_atom, int32_t* row_channel,
                      int capacity);
size_t mtr_head_packed_bytes(int C, int J, int D, int feat_dtype);
int mtr_head_pack_weights(const float* weight /*[J*(1+D), C] f32*/, const float* bias /*[J*(1+D)]*/,
                          int C, int J, int D, int feat_dtype, void* packed, mtr_stream_t stream);
int mtr_head_fused(const void* features, int feat_dtype, int layout, int B, int C, int H, int W,
                   const void* packed, int J, int D, const mtr_head_params* p, float* coords2d,
                   float* coords3d_rel, mtr_stream_t stream);
int mtr_head_fused_opts(const void* features, int feat_dtype, int layout, int B, int C, int H, int W,
                        const void* packed, int J, int D, const mtr_head_params* p,
                        const mtr_head_options* options, float* coords2d, float* coords3d_rel,
                        mtr_stream_t stream);
/* host-only, no GPU work: which kernel mtr_head_fused_ws takes for a launch (what a profile will show).
 * kernel: one of MTR_HEAD_KERNEL_*; 0 with return MTR_E_SHAPE = no fused kernel for this shape. */
enum {
  MTR_HEAD_KERNEL_RT = 1,        /* head_rt_kernel: 256 threads, every wave copies and multiplies          */
  MTR_HEAD_KERNEL_RT_LOADER = 2, /* head_rt_ld_kernel: four MFMA waves + a loader wave                     */
  MTR_HEAD_KERNEL_RT_KS = 3,     /* head_rt_ks_kernel: two K groups                                        */
  MTR_HEAD_KERNEL_RT_NP = 4,     /* head_rt_np_kernel: row tiles x column blocks per workgroup             */
  MTR_HEAD_KERNEL_16 = 10,       /* head_fused16_kernel: 16-bit features staged through registers          */
  MTR_HEAD_KERNEL_16_DMA = 11,   /* head_fused16dma_kernel: 16-bit features staged by global_load_lds      */
  MTR_HEAD_KERNEL_16_DMA_EARLY = 14,  /* head_fused16dma_kernel<..., EARLY>: copies of stage s + 1 issued first in stage s */
  MTR_HEAD_KERNEL_16_DMA_LOADER = 13, /* head_fused16dma_kernel<..., LD>: the same with a loader wave (dma_staging 2) */
  MTR_HEAD_KERNEL_16_AREG = 15,  /* head_fused16areg_kernel: weights in registers, a wave per joint group (dma_staging 4;
                                    the library's choice for >= 512 crops of >= 8 joint groups on 5 column tiles)  */
  MTR_HEAD_KERNEL_16_RES = 16,   /* head_fused16res_kernel: weights RESIDENT in registers, persistent workgroups (dma_staging 5) */
  MTR_HEAD_KERNEL_16_PP = 17,    /* head_fused16pp_kernel: eight waves in two alternating halves, four joint groups per
                                    workgroup (dma_staging 6; round 6)                                        */
  MTR_HEAD_KERNEL_16_DMA_EARLY_TIGHT = 18, /* the early-copies kernel on a feature stage of exactly H*W positions and two stages
                                    of LDS: one joint group per workgroup = 52 KiB at 12x12, three workgroups per CU (round 6;
                                    dma_staging 7; the library's choice where its round model says so)          */
  MTR_HEAD_KERNEL_16_RT = 12     /* head_rt16_kernel: 16-bit features on the row-tile core (1 + D > 64 rows per
                                    joint, or maps of more than 256 positions); NCHW features: needs the
                                    workspace (one transposing pass in front)                              */
};
typedef struct mtr_head_plan_info {
  int32_t kernel;
  int32_t tiles_per_workgroup;    /* f32: row tiles; 16-bit: joint groups                                   */
  int32_t column_blocks;          /* f32 np kernel: column blocks per workgroup tile, else 1               */
  int32_t split_column_blocks;    /* f32: column blocks dealt to workgroups (+ the merge launch), else 0   */
  int64_t workgroups;             /* of the main launch                                                     */
  double model_us;                /* f32 row-tile kernels: the launch plan's own estimate of the launch time
                                     (its cost model simulates the launch; diagnostic), else 0              */
} mtr_head_plan_info;
int mtr_head_plan(int feat_dtype, int layout, int B, int C, int H, int W, int J, int D,
                  const mtr_head_options* options, int have_workspace, mtr_head_plan_info* plan);

/* The same with a caller-provided scratch buffer (8-byte aligned, mtr_head_workspace_bytes(...) bytes,
 * contents irrelevant before and after; 0 bytes = this shape never uses one): with it, f32 maps of
 * more than 64 positions may spread their column blocks over workgroups (rt_split_column_blocks).
 * workspace == NULL is mtr_head_fused_opts.  Still no allocation, no sync, no globals: two launches
 * on `stream` instead of one when the split is taken. */
size_t mtr_head_workspace_bytes(int feat_dtype, int layout, int B, int C, int H, int W, int J, int D);
int mtr_head_fused_ws(const void* features, int feat_dtype, int layout, int B, int C, int H, int W,
                      const void* packed, int J, int D, const mtr_head_params* p,
                      const mtr_head_options* options, void* workspace, size_t workspace_bytes,
                      float* coords2d, float* coords3d_rel, mtr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K5: absolute (camera-space) reconstruction.  Replaces
 *   ptu3d.reconstruct_absolute        metrabs_pytorch/ptu3d.py:9-33
 *   ptu3d.reconstruct_ref_fullpersp   metrabs_pytorch/ptu3d.py:56-105
 *   ptu3d.reconstruct_ref_weakpersp   metrabs_pytorch/ptu3d.py:36-49
 *   ptu3d.is_within_fov / back_project           ptu3d.py:108-121
 *
 * The two RMS scalars of reconstruct_ref_fullpersp are taken over the WHOLE call batch
 * (ptu3d.py:71-74): B here must be the reference's internal batch for bit-comparable results.
 * workspace: mtr_reconstruct_workspace_bytes(B, J) bytes, 16-byte aligned.
 *
 * The split form exposes the batch moments so that a sharded caller can all-reduce them
 * (SURVEY.md section 8e "exact-monolithic mode"):
 *   mtr_reconstruct_moments -> moments[0] = sum(normalized2d^2), moments[1] = sum(rel_backproj^2),
 *                              moments[2] = number of summed elements (B*J*2), all f64;
 *   mtr_reconstruct_solve   <- the (possibly all-reduced) moments.
 */
size_t mtr_reconstruct_workspace_bytes(int B, int J);
int mtr_reconstruct_absolute(const float* coords2d /*[B,J,2]*/, const float* coords3d_rel /*[B,J,3]*/,
                             const float* intrinsics /*[B,3,3]*/, int B, int J,
                             const mtr_recon_params* p, float* poses3d /*[B,J,3]*/, void* workspace,
                             size_t workspace_bytes, mtr_stream_t stream);
int mtr_reconstruct_moments(const float* coords2d, const float* coords3d_rel,
                            const float* intrinsics, int B, int J, double* moments /*[3]*/,
                            void* workspace, size_t workspace_bytes, mtr_stream_t stream);
int mtr_reconstruct_solve(const float* coords2d, const float* coords3d_rel, const float* intrinsics,
                          int B, int J, const mtr_recon_params* p, const double* moments /*[3]*/,
                          float* poses3d, mtr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K6: crop sampler.
 *
 * mtr_build_pyramid replaces the whole-image gamma decode `(u8/255)**2.2`
 * (multiperson_model.py:196) and the box-filter pyramid (warping.py:10-13):
 *   images_u8 [N,3,Hi,Wi] u8  ->  level0 [N,3,Hi,Wi], level1 [N,3,Hi/2,Wi/2], level2 [N,3,Hi/4,Wi/4]
 *   (f32 linear light, floor division on odd sizes as avg_pool2d(2,2)).
 *
 * mtr_crop_geometry replaces Pose3dEstimator._get_new_rotation_and_scale and the matrix set-up of
 * _get_crops (multiperson_model.py:264-305,322-355; ptu3d.lookat_matrix ptu3d.py:129-142;
 * warping.undistort_points warping.py:65-73): one thread per (aug, box).
 *   boxes [n_box,4(+)] (x,y,w,h; row stride box_stride floats), intrinsics [n_box,3,3],
 *   distortion [n_box,12] (zero padded), camspace_up [n_box,3], aug_rotflipmat [n_aug,3,3],
 *   aug_scales [n_aug]
 *   -> new_intrinsics [n_aug,n_box,3,3], rot [n_aug,n_box,3,3], warp_params [n_aug*n_box, MTR_WARP_PARAM_FLOATS].
 *
 * mtr_warp_crops replaces warping.warp_images_with_pyramid / warp_single_image (warping.py:6-54:
 * homography, lens distortion warping.py:57-107, per-level intrinsics warping.py:128-133,
 * grid_sample bilinear/zeros/align_corners=True), the antialias avg_pool2d and the per-crop gamma
 * `crops **= gamma/2.2` of _get_crops (multiperson_model.py:308-319).
 *   warp_params [n_crops, MTR_WARP_PARAM_FLOATS] as laid out below; out [n_crops,3,res,res]
 *   (NCHW) or [n_crops,res,res,3] (NHWC), dtype f32/f16/bf16.
 */
#define MTR_WARP_PARAM_FLOATS 36
/* warp_params row (floats): [0..8] Hinv (new_invprojmat incl. the antialias scale), [9..17] K of
 * the chosen pyramid level, [18..29] 12 distortion coefficients, [30] has_distortion (0/1),
 * [31] pyramid level (0..2), [32] image id, [33] gamma exponent (gamma_aug / 2.2), [34..35] pad */

int mtr_build_pyramid(const uint8_t* images_u8, int N, int Hi, int Wi, float* level0, float* level1,
                      float* level2, mtr_stream_t stream);
/* Same pyramid from an already linear-light f32 level 0 (the argument warping.warp_images_with_pyramid
 * receives, warping.py:6-13): writes level1 / level2 only. */
int mtr_pyramid_from_level0(const float* level0, int N, int Hi, int Wi, float* level1, float* level2,
                            mtr_stream_t stream);
/* Level 0 left as the uint8 frame: the f32 level 0 is 64 % of the pyramid's bytes and is only ever
 * sampled, so the fast path never materialises it.  mtr_build_pyramid_u8 writes level1 / level2 and
 * the 256-entry gamma LUT (lut[v] = (v/255)**2.2); mtr_warp_crops_u8 samples level 0 from
 * images_u8 through that LUT (bit-identical to mtr_warp_crops on the materialised level 0). */
int mtr_build_pyramid_u8(const uint8_t* images_u8, int N, int Hi, int Wi, float* lut /*[256]*/,
                         float* level1, float* level2, mtr_stream_t stream);
int mtr_warp_crops_u8(const uint8_t* level0_u8, const float* lut, const float* level1,
                      const float* level2, int N, int Hi, int Wi, const float* warp_params,
                      int n_crops, int res, int antialias, int out_dtype, int out_layout, void* out,
                      mtr_stream_t stream);
/* The same two calls for frames with interleaved channels, images_u8 [N,Hi,Wi,3] (the memory of a
 * torch channels_last [N,3,Hi,Wi] tensor; what image decoders and numpy hand over; the reference
 * accepts such a tensor through its strides, `images.float()` at multiperson_model.py:196).  Levels 1
 * and 2 are the same f32 planes; the crops are bit-identical to mtr_warp_crops_u8 on the planar copy
 * of the frames.  The two x-taps of a row are then 6 consecutive bytes for all three channels: one
 * 12-byte gather per row instead of three 8-byte ones. */
int mtr_build_pyramid_u8_hwc(const uint8_t* images_u8, int N, int Hi, int Wi, float* lut /*[256]*/,
                             float* level1, float* level2, mtr_stream_t stream);
int mtr_warp_crops_u8_hwc(const uint8_t* level0_u8, const float* lut, const float* level1,
                          const float* level2, int N, int Hi, int Wi, const float* warp_params,
                          int n_crops, int res, int antialias, int out_dtype, int out_layout, void* out,
                          mtr_stream_t stream);
int mtr_crop_geometry(const float* boxes, int box_stride, const float* intrinsics,
                      const float* distortion, const float* camspace_up, const int32_t* image_ids,
                      const float* aug_rotflipmat, const float* aug_scales, const float* aug_gammas,
                      int n_box, int n_aug, int res, int antialias, float* new_intrinsics,
                      float* rot, float* warp_params, mtr_stream_t stream);
int mtr_warp_crops(const float* level0, const float* level1, const float* level2, int N, int Hi,
                   int Wi, const float* warp_params, int n_crops, int res, int antialias,
                   int out_dtype, int out_layout, void* out, mtr_stream_t stream);
/* antialias_factor > 4 (multiperson_model.py:312-315): the crops are sampled at res*aa x res*aa
 * (mtr_warp_crops[_u8] with res = res*aa, antialias = 1 and gamma exponents of 1 in warp_params[33])
 * and shrunk here the way torchvision.transforms.functional.resize(antialias=True) = aten's
 * separable antialiased bilinear filter does, with the per-crop `** (gamma / 2.2)` (:319) on the way
 * out.  crops_big [n_crops,3,res*aa,res*aa] f32 linear light; warp_params: the ORIGINAL rows (their
 * [33] is the exponent applied); workspace: mtr_crops_shrink_workspace_bytes(...) bytes;
 * 2 <= antialias <= 19. */
size_t mtr_crops_shrink_workspace_bytes(int n_crops, int res, int antialias);
int mtr_crops_shrink_antialiased(const float* crops_big, const float* warp_params, int n_crops, int res,
                                 int antialias, int out_dtype, int out_layout, void* out,
                                 void* workspace, size_t workspace_bytes, mtr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K7 (first "next" row): post-processing of the crop-model output in one launch.
 * Replaces Pose3dEstimator._predict_single_batch post-ops (multiperson_model.py:244-259: mirror
 * un-swap through joint_info.mirror_mapping for flipped augs, poses @ R, transpose) and the
 * post-ops of _estimate_poses_batched (multiperson_model.py:143-178: joint_transform_matrix,
 * 2D projection with lens distortion, world transform by inv(extrinsics), skeleton selection,
 * mean over the TTA axis).
 *   poses_crop [A, n, J, 3] crop-model output (aug-major), rot [A, n, 3, 3], should_flip [A] (u8),
 *   mirror_mapping [J] (i32), joint_transform [J, Jt] or NULL, skeleton [S] (i32) or NULL,
 *   intrinsics [n,3,3], distortion [n,12], inv_extrinsics [n,4,4] (already inverted)
 *   -> poses3d [n, (A,) S', 3], poses2d [n, (A,) S', 2]; the A axis is dropped when average_aug;
 *   S' = S if skeleton else (Jt if joint_transform else J).
 */
int mtr_postprocess_poses(const float* poses_crop, const float* rot, const uint8_t* should_flip,
                          const int32_t* mirror_mapping, const float* joint_transform, int Jt,
                          const int32_t* skeleton, int S, const float* intrinsics,
                          const float* distortion, const float* inv_extrinsics, int A, int n, int J,
                          int average_aug, float* poses3d, float* poses2d, mtr_stream_t stream);

/* Row a11: Metrabs.latent_points_to_joints (metrabs_tf/models/metrabs.py:80-81 ->
 * tfu3d.linear_combine_points tfu3d.py:48-49, einsum 'bjc,jJ->bJc'), the step Metrabs.forward runs
 * behind reconstruct_absolute when transform_coords / predict_all_and_latents is set
 * (metrabs_pytorch/models/metrabs.py:61-62, metrabs_tf/models/metrabs.py:61-62).
 *   points [B, J_in, 3] f32 (the reconstructed latent points), weights [J_in, J_out] f32 row-major
 *   (`w2` of the affine-weights file = recombination_weights) -> out [B, J_out, 3] f32.
 * f64 sums, one rounding per output.  J_in <= 4096. */
int mtr_linear_combine_points(const float* points, const float* weights, int B, int J_in, int J_out,
                              float* out, mtr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K9: detector pre-processing, the step in front of the hot path.
 * Replaces PersonDetector.forward minus the network (metrabs_pytorch/multiperson/person_detector.py):
 *   :15-20,26-29  target size / padding arithmetic in float32        -> mtr_detector_geometry (host)
 *   :21-33        (u8/255)**2.2, torchvision bilinear resize (antialiased when shrinking),
 *                 **(1/2.2), pad to multiples of 32 with 0.5           -> mtr_detector_preprocess
 *   :47-54        scale_boxes (network frame -> image frame)          -> mtr_detector_scale_boxes
 * The resize mirrors aten's separable CPU kernels operation for operation (bit-exact in linear
 * light); shrink factors above 19x are rejected (MTR_E_SHAPE).
 */
typedef struct mtr_detector_geom {
  int32_t target_h, target_w; /* resized extent: int32(factor * h), int32(factor * w)          */
  int32_t antialias;          /* factor < 1                                                    */
  int32_t pad_top, pad_left;  /* half_pad_h, half_pad_w                                        */
  int32_t out_h, out_w;       /* padded extent (multiples of 32)                               */
  float x_factor, y_factor;   /* w / target_w, h / target_h                                    */
} mtr_detector_geom;

/* host-only, no GPU work: fills `g` (a HOST pointer) for frames of H x W */
int mtr_detector_geometry(int H, int W, int input_size /*416*/, mtr_detector_geom* g);
/* images_u8 [N,3,H,W] uint8 -> out [N,3,g->out_h,g->out_w] f32 (what the network is fed); g: host */
int mtr_detector_preprocess(const uint8_t* images_u8, int N, int H, int W, const mtr_detector_geom* g,
                            float* out, mtr_stream_t stream);
/* The same with the kernel named (tests and timing; the two kernels give identical bits):
 *   AUTO    the streaming kernel for launches past its measured cross-over (4 frames of 1080p; any
 *           shrink of 5.5x and more) when the frame tensor is a multiple of 16 bytes and its LDS fits,
 *           else the tile kernel (what mtr_detector_preprocess does);
 *   TILE    one 8-row x 64-column output tile at a time, staged through registers;
 *   STREAM  a 64-column strip walked down the frame, rows loaded global -> LDS two chunks ahead by
 *           a loading wave (MTR_E_SHAPE when not available for this tensor). */
enum { MTR_DETECTOR_KERNEL_AUTO = 0, MTR_DETECTOR_KERNEL_TILE = 1, MTR_DETECTOR_KERNEL_STREAM = 2 };
int mtr_detector_preprocess_kernel(const uint8_t* images_u8, int N, int H, int W, const mtr_detector_geom* g,
                                   int kernel, float* out, mtr_stream_t stream);
/* xyxy_conf [n,5] (x1,y1,x2,y2,conf; network frame) -> boxes_out [n,5] (x,y,w,h,conf; image frame) */
int mtr_detector_scale_boxes(const float* xyxy_conf, int n, const mtr_detector_geom* g,
                             float* boxes_out, mtr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K8: plausibility filter + pose NMS, one launch per call (one
 * workgroup per image).  Replaces _filter_poses (metrabs_tf/multiperson/multiperson_model.py:441-459)
 * and plausibility_check.py (TF :9-96; PyTorch port metrabs_pytorch/multiperson/
 * plausibility_check.py:8-119, whose call site multiperson_model.py:158-163 is commented out):
 * is_pose_plausible, are_augmentation_results_consistent, is_pose_consistent_with_box,
 * compute_pose_similarity, pose_non_max_suppression.
 *   poses3d [P,A,J,3] camera-space mm (all model joints, before the skeleton selection),
 *   poses2d [P,A,J,2] px, boxes [P,5] (x,y,w,h,score); poses of image i are rows
 *   row_start[i] .. row_start[i+1]-1; sim_offsets[i] = sum_{k<i} n_k^2 (floats);
 *   edges [n_bones,2] + mean_bones [n_bones] of the first J_model joints (n_bones = 0: no bone test).
 *   -> valid [P] (u8, the three plausibility tests), keep_idx [P] (per image: the kept poses as
 *      global row indices, then -1), keep_count [n_images].
 */
typedef struct mtr_filter_params {
  float rel_small, rel_big, abs_diff_mm; /* 0.1, 3, 300   plausibility_check.py:22-24            */
  float stdev_mm;                        /* 200           :64-66                                 */
  float box_fraction;                    /* 0.5           TF :84                                  */
  float sim_scale_mm, sim_threshold;     /* 300, 0.4      :83, :59                                */
  int32_t max_output;                    /* 150 (TF max_output_size; used when order_by_score)     */
  int32_t var_correction;                /* 0 = population variance (TF), 1 = torch.var default    */
  int32_t order_by_score;                /* 1 = TF order (score descending), 0 = PyTorch port (index) */
} mtr_filter_params;

size_t mtr_filter_poses_workspace_bytes(int P, int J, int max_per_image);
int mtr_filter_poses(const float* poses3d, const float* poses2d, const float* boxes,
                     const int32_t* row_start, const int32_t* sim_offsets, int n_images, int P, int A,
                     int J, const int32_t* edges, const float* mean_bones, int n_bones, int J_model,
                     const mtr_filter_params* params /*host*/, void* workspace, size_t workspace_bytes,
                     int max_per_image, uint8_t* valid, int32_t* keep_idx, int32_t* keep_count,
                     mtr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K10 (outside the reference's hot path): fused epilogue for the PyTorch-ROCm
 * backbone's inference copy, metrabs_amd/backbones.py:fold_batchnorm(fused_epilogue=True).  With
 * batch norm folded into a convolution, what follows it is "+ bias[c]" and an activation -- two
 * elementwise kernels in PyTorch-ROCm (torchvision Conv2dNormActivation: ops/misc.py; the
 * reference's backbones/efficientnet.py:11-18 builds on it).  In place on y [B, C, HW] (NCHW,
 * 16-byte aligned, HW % (16 / sizeof(dtype)) == 0): y = act(y + bias[c]) (+ residual, the skip
 * connection of an (Fused)MBConv block, same shape and dtype as y, may be NULL), computed in f32.
 * act: 0 none, 1 ReLU, 2 SiLU, 3 Hardswish.
 */
int mtr_bias_act_nchw(void* y, int dtype, const float* bias /*[C] f32*/, const void* residual, int act,
                      long long B, int C, int HW, mtr_stream_t stream);

/* Same pass without a residual, additionally row_mean[b * C + c] = mean over H*W of the stored result:
 * the input of the squeeze-excite block behind a depthwise convolution (x.mean((2, 3)), the first
 * statement of torchvision's SqueezeExcitation that efficientnet.py:110-173 uses), otherwise a
 * reduction kernel of its own. */
int mtr_bias_act_rowmean_nchw(void* y, int dtype, const float* bias /*[C] f32*/, int act, long long B,
                              int C, int HW, float* row_mean /*[B*C] f32*/, mtr_stream_t stream);

/* K11 (outside the reference's hot path, like K10): depthwise 3x3 convolution of the backbone's
 * inference copy with the K10 epilogue in the same pass.  x [B, C, H, W] -> y [B, C, OH, OW] (NCHW,
 * same dtype; OH = (H + 2 pad - 3) / stride + 1, OW likewise and a multiple of 4), cross-correlation
 * with weight [C, 3, 3] f32 (torch Conv2d(groups=C).weight with the batch norm folded in), zero
 * padding `pad` in {0, 1} on every side, stride in {1, 2}; y = act(conv + bias[c]) in f32 arithmetic;
 * row_mean (may be NULL): [B*C] f32 mean of the stored result per (b, c) plane, the input of the
 * squeeze-excite block behind the layer (efficientnet.py:110-173). */
int mtr_depthwise3x3_bias_act(const void* x, int dtype, const float* weight, const float* bias, int act,
                              long long B, int C, int H, int W, int stride, int pad, void* y,
                              float* row_mean, mtr_stream_t stream);
/* The same layer with the explicit asymmetric zero padding the reference's TF-'SAME' stride-2 layers
 * put in front of an unpadded convolution (efficientnet.py:1127-1161: (0,1,0,1), or (0,2,0,2) for the
 * `bottomright_stride` layer) folded in: pad_top, pad_left in {0, 1}, pad_bottom, pad_right in 0..2;
 * OH = (H + pad_top + pad_bottom - 3) / stride + 1, OW likewise and a multiple of 4.  x is the
 * UNPADDED activation: the ZeroPad2d pass and its padded copy disappear. */
int mtr_depthwise3x3_bias_act_padded(const void* x, int dtype, const float* weight, const float* bias,
                                     int act, long long B, int C, int H, int W, int stride, int pad_top,
                                     int pad_left, int pad_bottom, int pad_right, void* y,
                                     float* row_mean, mtr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* METRABS_HIP_H_ */
