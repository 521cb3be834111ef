// K1 + K2-K4 fused: 1x1 heatmap projection on the matrix cores with the volumetric soft-argmax
// decode as the epilogue -- logits never reach HBM.
//
// Replaces MetrabsHeads.forward as a whole (metrabs_pytorch/models/metrabs.py:75-85):
//   x = conv_final(inp)                               (LazyConv2d 1x1, models/metrabs.py:73,76)
//   split / rearrange 'b (d j) h w -> b d j h w'      (:78-79)
//   soft_argmax 3D + heatmap_to_metric, soft_argmax 2D + heatmap_to_image   (:80-83)
//
// Per crop the projection is a GEMM  logits[N_out, HW] = Wt[N_out, C] . feat[C, HW]  (N_out =
// J*(1+D) = 153, HW = 64, C = 1280 for EffNetV2-S/256): 25 MFLOP over 328 KB of features
// = 76 FLOP/B, above the f32-MFMA ridge (157 TF / 6.3-8 TB/s = 20-25), so in fp32 this kernel is
// MATRIX-bound.  Precision class follows the feature dtype:
//   * f32 features (the reference's CPU path): f32-input MFMA over SHORT chains (16 channels),
//     each chain's result carried into f64 accumulators on the VALU.  Why not a
//     plain f32 chain: over K = 1280 it is ~4x noisier than oneDNN's blocked accumulation (1.6e-3
//     vs 3.7e-4 mm from the fp64 truth on the golden cases) and fails the 1e-3 mm gate.  Why not
//     v_mfma_f64_16x16x4_f64 (round-1 first choice, exact products + f64 accumulate): a pure chain of
//     it, operands in registers, measures 47.7 TF at 4 waves/SIMD and 33-35 TF at 1 wave/SIMD on
//     this chip (tools/experiments/mfma_probe.hip) -- 61 % / 42 % of the 78.6 TF spec -- and the
//     kernel already sat at 90-95 % of that ceiling (45 TF), whereas the f32 shape reaches 125 TF;
//   * f16 / bf16 features (the autocast GPU path, where the reference itself rounds the logits to
//     f16): one f32 fma chain over all of K; weights and logits stay f32, i.e. strictly more
//     accurate than the reference's f16 logits.
//
// Two GEMM cores share the packing, the grid mapping and the decode epilogue: the 16x16x4 core
// described next (maps of <= 32 or 129..256 positions) and the 32x32x2 core further down (33..128
// positions, i.e. the 8x8 maps of the 256-px models), which is the faster one where it applies.
//
// Decomposition (16x16x4 core)
//   * weights are re-packed once (mtr_head_pack_weights) joint-major: joint j owns rows
//     [2D chan j, depth 0 .. D-1] so a JOINT GROUP is a contiguous <=64-row block that can be
//     decoded without leaving the workgroup; layout [group][c][64 rows] (k-major) so that the
//     weight tile is staged with the same full-line 16-B loads as the feature tile;
//   * one workgroup (4 waves) = (crop, joint group): wave w owns row tile w (16 rows) x all NT
//     column tiles; K streams through LDS in 32-channel stages, double buffered, global loads for
//     stage s+1 in flight under the MFMAs of stage s;
//   * LDS rows are padded so that row stride = 16 (mod 32) words: the A/B fragment reads
//     (lane (l&15, l>>4) -> [k0 + (l>>4)][16*tile + (l&15)]) are bank-conflict free;
//   * epilogue: accumulators (+bias) -> LDS [64][HWpad], then each half-wave decodes one joint
//     (softmax over its D slices, fp64 moment sums; exp in the accumulator's precision class);
//   * 1-D grid with an XCD-aware remap: the joint groups of one crop run on the same XCD so the
//     crop's features are fetched from HBM once and re-read from that XCD's L2.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"
#include "head_rt.h"

namespace mtr {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;

constexpr int kRows = 64;       // rows (output channels) per workgroup = 4 waves x 16
constexpr int kRowsPad = 80;    // LDS row stride of the weight tile, 80 = 16 (mod 32)
constexpr int kKC = 32;         // channels per pipeline stage

__host__ __device__ constexpr int hw_pad(int nt) { return (nt & 1) ? nt * 16 : nt * 16 + 16; }

struct HeadGeom {
  int n_groups;      // joint groups
  int jg;            // joints per group (last group may hold fewer)
  int c_pad;         // C rounded up to kKC
};

__host__ __device__ inline HeadGeom head_geom(int C, int J, int D) {
  HeadGeom g;
  const int per = 1 + D;
  const int jg_max = kRows / per;  // >= 1 is checked by the caller
  g.n_groups = (J + jg_max - 1) / jg_max;
  g.jg = (J + g.n_groups - 1) / g.n_groups;  // balanced groups
  g.c_pad = (C + kKC - 1) / kKC * kKC;
  return g;
}

// packed = [n_groups][c_pad][64] weights (16x16 core), [n_groups][64] bias,
//          [n_groups][c_pad/32][64][32] weights (32x32 core)  -- all f32
__global__ void head_pack_kernel(const float* __restrict__ w, const float* __restrict__ bias, int C,
                                 int J, int D, HeadGeom g, float* __restrict__ packed) {
  const int per = 1 + D;
  const size_t n_w = (size_t)g.n_groups * g.c_pad * kRows;
  const size_t n_b = (size_t)g.n_groups * kRows;
  const size_t total = 2 * n_w + n_b;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (size_t)gridDim.x * blockDim.x) {
    int row, c, grp;
    const bool is_bias = t >= n_w && t < n_w + n_b;
    if (t < n_w) {
      row = (int)(t % kRows);
      c = (int)((t / kRows) % g.c_pad);
      grp = (int)(t / ((size_t)kRows * g.c_pad));
    } else if (is_bias) {
      row = (int)((t - n_w) % kRows);
      c = 0;
      grp = (int)((t - n_w) / kRows);
    } else {
      const size_t u = t - n_w - n_b;
      row = (int)((u / kKC) % kRows);
      const size_t st = u / ((size_t)kKC * kRows);  // global stage index = grp * n_stages + stage
      grp = (int)(st / (g.c_pad / kKC));
      c = (int)(st % (g.c_pad / kKC)) * kKC + (int)(u % kKC);
    }
    const int jl = row / per, k = row % per;
    const int j = grp * g.jg + jl;
    float v = 0.0f;
    if (jl < g.jg && j < J && c < C) {
      // reference channel order: n = j for the 2D map, J + d*J + j for depth slice d
      const int n = (k == 0) ? j : J + (k - 1) * J + j;
      v = is_bias ? bias[n] : w[(size_t)n * C + c];
    }
    packed[t] = v;
  }
}

// ---- global -> registers -> LDS staging of one 32-channel stage (weights tile + feature tile).
// Native ext_vector loads/stores only: copying HIP's float4 *struct* between address spaces lowers
// to llvm.memcpy (global -> private -> LDS), which SROA does not split, so the staged tile went
// through scratch memory with a dependent scratch_load -> ds_write -> barrier chain every stage.
using v4f = __attribute__((ext_vector_type(4))) float;

template <typename T>
__device__ __forceinline__ v4f load4_native(const T* p) {
  float v[4];
  load_vec<T, 4>(p, v);
  return v4f{v[0], v[1], v[2], v[3]};
}
template <>
__device__ __forceinline__ v4f load4_native<float>(const float* p) {
  return *reinterpret_cast<const v4f*>(p);
}

template <int B_VECS>
struct StageRegs {
  v4f a[2];
  v4f b[B_VECS];
};
template <typename FeatT>
struct StageSrc {
  const float* wgrp;    // [c_pad][64] packed weights of this joint group
  const FeatT* fcrop;   // [C][HW] features of this crop
  int C, HW, vec_per_row, b_total, tid;
};

constexpr int kKP = 34;  // NHWC feature tile [position][34]: row stride = 2 (mod 32) words

// NCHW: the stage is kKC rows (channels) of HW contiguous values -> LDS [k][HWP].
// NHWC (torch channels_last / the TF twin, tf models/metrabs.py:100-101): per position kKC
// contiguous channels (128 B) -> LDS [position][kKP]; fragment reads (lane (j, g) -> [n*16+j][k0+g])
// hit banks 2j+g: conflict-free.
template <typename FeatT, int B_VECS, bool NHWC>
__device__ __forceinline__ void load_stage(const StageSrc<FeatT>& s, int c0, StageRegs<B_VECS>& r) {
  // weight tile: rows c0..c0+31 of [c_pad][64], fully contiguous 8 KiB
#pragma unroll
  for (int i = 0; i < 2; ++i)
    r.a[i] = *reinterpret_cast<const v4f*>(s.wgrp + (size_t)c0 * kRows + (size_t)(s.tid + i * 256) * 4);
#pragma unroll
  for (int i = 0; i < B_VECS; ++i) {
    const int v = s.tid + i * 256;
    bool ok;
    size_t off;
    if constexpr (NHWC) {
      const int pos = v / (kKC / 4), c4 = v % (kKC / 4);
      ok = pos < s.HW && c0 + c4 * 4 < s.C;
      off = (size_t)pos * s.C + c0 + c4 * 4;
    } else {
      const int row = v / s.vec_per_row, q = v - row * s.vec_per_row;
      ok = v < s.b_total && c0 + row < s.C;
      off = (size_t)(c0 + row) * s.HW + q * 4;
    }
    // clamp instead of branching: every lane loads a valid address, invalid lanes get zeros
    const v4f val = load4_native<FeatT>(s.fcrop + (ok ? off : 0));
    r.b[i] = ok ? val : v4f{0.f, 0.f, 0.f, 0.f};
  }
}

template <int B_VECS, int HWP, bool NHWC, typename FeatT>
__device__ __forceinline__ void store_stage(const StageSrc<FeatT>& s, float* As_buf, float* Bs_buf,
                                            const StageRegs<B_VECS>& r) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int v = s.tid + i * 256;
    const int row = v / (kRows / 4), q = v % (kRows / 4);
    *reinterpret_cast<v4f*>(As_buf + row * kRowsPad + q * 4) = r.a[i];
  }
#pragma unroll
  for (int i = 0; i < B_VECS; ++i) {
    const int v = s.tid + i * 256;
    if constexpr (NHWC) {
      using v2f = __attribute__((ext_vector_type(2))) float;
      const int pos = v / (kKC / 4), c4 = v % (kKC / 4);
      if (pos < s.HW) {  // rows of 136 B: 8-byte aligned -> two 8-byte writes
        float* dst = Bs_buf + pos * kKP + c4 * 4;
        *reinterpret_cast<v2f*>(dst) = v2f{r.b[i][0], r.b[i][1]};
        *reinterpret_cast<v2f*>(dst + 2) = v2f{r.b[i][2], r.b[i][3]};
      }
    } else {
      const int row = v / s.vec_per_row, q = v - row * s.vec_per_row;
      if (v < s.b_total) *reinterpret_cast<v4f*>(Bs_buf + row * HWP + q * 4) = r.b[i];
    }
  }
}

// ---- decode epilogue shared by both GEMM kernels: logits of one joint group in LDS [64][HWP]
// (row = jl*(1+D) + {0: 2D map, 1+d: depth slice d}); a half-wave (32 lanes) per joint (<= 8 joints
// in flight).  The logits are on chip and the epilogue is a few % of the GEMM, so the f64-accumulate
// mode also takes exp in f64: the decode error then is the f32 rounding of the outputs only, which
// matters because reconstruct_absolute amplifies coords3d_rel errors ~7x.
// PV = positions per lane and step: 4 for maps of more than 64 positions, 2 below (an 8x8 map
// then keeps all 32 lanes of the half-wave busy instead of 16).
template <bool ACC64, int PV>
__device__ __forceinline__ void decode_group_from_lds(const float* Ls, int HWP, int grp,
                                                      const HeadGeom& g, int crop, int J, int D,
                                                      int H, int W, const HeadScale& hs,
                                                      float* __restrict__ coords2d,
                                                      float* __restrict__ coords3d_rel, int wid,
                                                      int lane) {
  using vecf = __attribute__((ext_vector_type(PV))) float;
  const int HW = H * W;
  const int per = 1 + D;
  const int li = lane & 31;
  for (int jl = wid * 2 + (lane >> 5); jl < g.jg; jl += 8) {
    const int j = grp * g.jg + jl;
    if (j >= J) continue;
    const float* row2d = Ls + (size_t)(jl * per) * HWP;
    const float* row3d = row2d + HWP;
    float m2 = -INFINITY, m3 = -INFINITY;
    for (int p = li * PV; p < HW; p += 32 * PV) {
      const vecf v = *reinterpret_cast<const vecf*>(row2d + p);
#pragma unroll
      for (int q = 0; q < PV; ++q) m2 = fmaxf(m2, v[q]);
      for (int d = 0; d < D; ++d) {
        const vecf u = *reinterpret_cast<const vecf*>(row3d + (size_t)d * HWP + p);
#pragma unroll
        for (int q = 0; q < PV; ++q) m3 = fmaxf(m3, u[q]);
      }
    }
    m2 = group_max<32>(m2);
    m3 = group_max<32>(m3);
    double s2 = 0, sx2 = 0, sy2 = 0, s3 = 0, sx3 = 0, sy3 = 0, sz3 = 0;
    for (int p = li * PV; p < HW; p += 32 * PV) {
      const vecf v2 = *reinterpret_cast<const vecf*>(row2d + p);
      double col[PV];
#pragma unroll
      for (int q = 0; q < PV; ++q) col[q] = 0;
      for (int d = 0; d < D; ++d) {
        const vecf u3 = *reinterpret_cast<const vecf*>(row3d + (size_t)d * HWP + p);
#pragma unroll
        for (int q = 0; q < PV; ++q) {
          const double e = ACC64 ? exp_neg64((double)u3[q] - (double)m3) : (double)expf(u3[q] - m3);
          col[q] += e;
          sz3 += e * (double)d;
        }
      }
#pragma unroll
      for (int q = 0; q < PV; ++q) {
        const int h = (p + q) / W, w = (p + q) - h * W;  // narrow maps wrap more than once
        const double e2 = ACC64 ? exp_neg64((double)v2[q] - (double)m2) : (double)expf(v2[q] - m2);
        s2 += e2; sx2 += e2 * w; sy2 += e2 * h;
        s3 += col[q]; sx3 += col[q] * w; sy3 += col[q] * h;
      }
    }
    s2 = group_sum<32>(s2); sx2 = group_sum<32>(sx2); sy2 = group_sum<32>(sy2);
    s3 = group_sum<32>(s3); sx3 = group_sum<32>(sx3); sy3 = group_sum<32>(sy3);
    sz3 = group_sum<32>(sz3);
    if (li == 0) {
      const size_t o = (size_t)crop * J + j;
      coords2d[o * 2 + 0] = heatmap_to_px(axis_coord(sx2, s2, W), hs);
      coords2d[o * 2 + 1] = heatmap_to_px(axis_coord(sy2, s2, H), hs);
      coords3d_rel[o * 3 + 0] = heatmap_to_mm_xy(axis_coord(sx3, s3, W), hs);
      coords3d_rel[o * 3 + 1] = heatmap_to_mm_xy(axis_coord(sy3, s3, H), hs);
      coords3d_rel[o * 3 + 2] = heatmap_to_mm_z(axis_coord(sz3, s3, D), hs);
    }
  }
}

// LDS (40-90 KiB per workgroup) already caps residency at <= 4 waves per SIMD; asking for 2 lets the
// register allocator keep the prefetched stage and the fragment batch in VGPRs instead of scratch.
template <typename FeatT, int NT, bool ACC64, bool NHWC>
__global__ __launch_bounds__(256, 2) void head_fused_kernel(
    const FeatT* __restrict__ feat, const float* __restrict__ packed, int B, int C, int H, int W,
    int J, int D, HeadGeom g, HeadScale hs, float* __restrict__ coords2d,
    float* __restrict__ coords3d_rel) {
  constexpr int HWP = hw_pad(NT);
  constexpr int A_STAGE = kKC * kRowsPad;            // floats
  constexpr int B_STAGE = NHWC ? NT * 16 * kKP : kKC * HWP;  // floats
  constexpr int B_VECS = (kKC * NT * 16 / 4 + 255) / 256;  // float4 per thread per stage (upper bound)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][kKC][kRowsPad]
  float* Bs = smem + 2 * A_STAGE;   // [2][kKC][HWP]
  float* Ls = smem;                 // epilogue alias: [kRows][HWP]

  const int HW = H * W;
  // ---- XCD-aware remap (block id b runs on XCD b % 8): the groups of a crop share an XCD
  const int chunk = 8 * g.n_groups;
  const int id = blockIdx.x;
  const int crop = (id / chunk) * 8 + (id % 8);
  const int grp = (id % chunk) / 8;
  if (crop >= B) return;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const FeatT* fcrop = feat + (size_t)crop * C * HW;
  const float* wgrp = packed + (size_t)grp * g.c_pad * kRows;
  const float* bgrp = packed + (size_t)g.n_groups * g.c_pad * kRows + (size_t)grp * kRows;

  const int vec_per_row = HW / 4;              // HW % 4 == 0 (checked on the host)
  const int b_total = kKC * vec_per_row;       // feature float4s per stage
  const StageSrc<FeatT> src{wgrp, fcrop, C, HW, vec_per_row, b_total, tid};

  using AccT = typename std::conditional<ACC64, f64x4, f32x4>::type;
  AccT acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = AccT{0, 0, 0, 0};

  // columns >= HW of the feature tile are never written: zero them once in both buffers
  if (NT * 16 > HW || NHWC) {  // (NHWC: also the 2 pad words of every row)
    for (int v = tid; v < 2 * B_STAGE; v += 256) Bs[v] = 0.0f;
    __syncthreads();
  }

  const int n_stages = g.c_pad / kKC;
  const int fr = lane & 15, fk = lane >> 4;

  // One pipeline iteration: MFMAs of stage s out of LDS buffer s&1, while the global loads of
  // stage s+2 are issued into register set LD and the (already landed) stage s+1 held in register
  // set ST is written to the other LDS buffer.  Two register sets alternate (HEAD_ITER is expanded
  // twice per loop trip so that both are statically indexed): two stages = 32 KiB per workgroup
  // are in flight, which is what it takes to cover the ~2 us load latency seen by PMC
  // (SQ_WAIT_ANY) when fewer than one workgroup per CU is resident (B = 64).
#define HEAD_ITER(S, LD, ST)                                                                      \
  {                                                                                               \
    const int s_ = (S);                                                                           \
    const int buf = s_ & 1;                                                                       \
    if (s_ + kAhead < n_stages) load_stage<FeatT, B_VECS, NHWC>(src, (s_ + kAhead) * kKC, LD);          \
    const float* Ab = As + buf * A_STAGE + wid * 16 + fr;                                         \
    const float* Bb = Bs + buf * B_STAGE + (NHWC ? fr * kKP : fr);                                \
    _Pragma("unroll") for (int kb = 0; kb < kKC / 4; kb += KS) {                                  \
      float af[KS], bf[KS][NT];                                                                   \
      _Pragma("unroll") for (int k = 0; k < KS; ++k) {                                            \
        af[k] = Ab[((kb + k) * 4 + fk) * kRowsPad];                                               \
        _Pragma("unroll") for (int n = 0; n < NT; ++n)                                            \
            bf[k][n] = NHWC ? Bb[n * 16 * kKP + (kb + k) * 4 + fk]                                \
                            : Bb[((kb + k) * 4 + fk) * HWP + n * 16];                             \
      }                                                                                           \
      __builtin_amdgcn_sched_barrier(0); /* keep the reads batched ahead of the MFMAs */          \
      if constexpr (ACC64) {                                                                      \
        /* f32 MFMA over a SHORT chain (KS k-steps = 4*KS channels), carried into f64 */          \
        f32x4 part[NT];                                                                           \
        _Pragma("unroll") for (int n = 0; n < NT; ++n) part[n] = f32x4{0.f, 0.f, 0.f, 0.f};       \
        _Pragma("unroll") for (int k = 0; k < KS; ++k) {                                          \
          _Pragma("unroll") for (int n = 0; n < NT; ++n)                                          \
              part[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k], bf[k][n], part[n], 0, 0, 0);  \
        }                                                                                         \
        _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                          \
          _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[n][r] += (double)part[n][r];          \
        }                                                                                         \
      } else {                                                                                    \
        _Pragma("unroll") for (int k = 0; k < KS; ++k) {                                          \
          _Pragma("unroll") for (int n = 0; n < NT; ++n)                                          \
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k], bf[k][n], acc[n], 0, 0, 0);    \
        }                                                                                         \
      }                                                                                           \
    }                                                                                             \
    if (s_ + 1 < n_stages)                                                                        \
      store_stage<B_VECS, HWP, NHWC>(src, As + (buf ^ 1) * A_STAGE, Bs + (buf ^ 1) * B_STAGE, ST);      \
    __syncthreads();                                                                              \
  }

  // k-steps per fragment batch (= the f32 chain length / 4 in carry mode) and prefetch depth are
  // register-budget choices: f64 carry accumulators take 8 VGPRs per tile (128 at NT = 16)
  // (carry interval 16 vs 32 channels: 32 is 4-6 % faster but 3x less accurate on peaked logits --
  //  1.6e-3 vs 4.9e-4 mm from fp64 on golden case s256_c1280_peaked; parity first)
  constexpr int KS = NT <= 4 ? (ACC64 ? 4 : 8) : (NT <= 9 ? 4 : 2);
  constexpr int kAhead = (NT >= 16 && ACC64) ? 1 : 2;
  StageRegs<B_VECS> regs0;
  load_stage<FeatT, B_VECS, NHWC>(src, 0, regs0);
  store_stage<B_VECS, HWP, NHWC>(src, As, Bs, regs0);
  if constexpr (kAhead == 2) {
    StageRegs<B_VECS> regs1;
    if (n_stages > 1) load_stage<FeatT, B_VECS, NHWC>(src, kKC, regs1);
    __syncthreads();
    for (int s = 0; s < n_stages; s += 2) {
      HEAD_ITER(s, regs0, regs1)
      if (s + 1 < n_stages) HEAD_ITER(s + 1, regs1, regs0)
    }
  } else {
    __syncthreads();
    for (int s = 0; s < n_stages; ++s) HEAD_ITER(s, regs0, regs0)
  }
#undef HEAD_ITER

  // ---- epilogue 1: logits (+bias) -> LDS [64][HWP].  C/D layout of f32 16x16x4: col = l&15,
  //   row = (l>>4)*4 + reg (the f64 carry accumulators mirror it element for element)
  // (the final __syncthreads of the loop already separates the last MFMA reads from these writes)
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wid * 16 + fk * 4 + r;
      // bias joins in the accumulator's precision; one rounding to f32
      Ls[row * HWP + n * 16 + fr] = (float)(acc[n][r] + (decltype(acc[n][r] + 0))bgrp[row]);
    }
  __syncthreads();

  decode_group_from_lds<ACC64, (NT > 4 ? 4 : 2)>(Ls, HWP, grp, g, crop, J, D, H, W, hs, coords2d, coords3d_rel, wid,
                               lane);
}

// =====================================================================================
// 32x32 variant (maps of 33..160 positions, i.e. every shipped configuration).
//
// Why a second GEMM core: with one workgroup per CU (B = 64 crops -> 256 workgroups) every SIMD
// holds ONE wave, and in that regime (a) a single wave gets about a fifth of the LDS rate on
// ds_read_b32 but the full rate on ds_read_b128 (MI355X_MICROARCH.md, LDS), and (b) a dependent
// v_mfma_f32_32x32x2_f32 chain runs at 149 TF from registers where 16x16x4 chains reach 101
// (tools/experiments/mfma_probe.hip).  So:
//   * both LDS tiles are K-CONTIGUOUS: weights [64 rows][32 ch], features [position][32 ch], one
//     128-byte row per output row / position and stage;
//   * the k index of the MFMA is a free permutation (A and B only have to agree): MFMA (u, s) of a
//     stage contracts channels {8u + s, 8u + 4 + s}; lane (i, g) therefore needs channels
//     8u + 4g .. + 3 of row i for s = 0..3 = ONE ds_read_b128 per operand per four MFMAs (was: one
//     ds_read_b32 per operand per MFMA);
//   * rows are XOR-swizzled in 16-byte slots, slot ^= swz(row), which makes those reads conflict-
//     free for the b128 lane groups {0-3,12-15,20-27},... and the transposing ds_write_b32 of NCHW
//     features at most 2-way (HW = 64; free) / 4-way;
//   * wave w owns row tile w & 1 and column tiles (w >> 1) + 2t; f32 chains of 16 channels
//     (8 MFMAs) are carried into f64 exactly as in the 16x16 core, from two alternating partial
//     sets so the VALU carry of one chain runs under the MFMAs of the next.
// Weights for this core are packed [group][stage][64 rows][32 ch] (appended to the 16x16 layout
// by mtr_head_pack_weights).
using f32x16 = __attribute__((ext_vector_type(16))) float;

// developer-only timing ablations of the 32x32 core (tools/experiments/ablate_head.sh); 0 in the product
#ifndef MTR_ABLATE
#define MTR_ABLATE 0
#endif

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 1); }

// Staged registers of the 32x32 core hold the bits as loaded (8 bytes for four f16 / bf16 values);
// the conversion to f32 happens at store time, next to the zeroing, for the same reason.
using v2u = __attribute__((ext_vector_type(2))) unsigned;
template <typename T> struct RawVec { using type = v2u; };
template <> struct RawVec<float> { using type = v4f; };

template <typename T>
__device__ __forceinline__ typename RawVec<T>::type load4_raw(const T* p) {
  return *reinterpret_cast<const typename RawVec<T>::type*>(p);
}
__device__ __forceinline__ v4f raw_to_f32(v4f v, const float*) { return v; }
template <typename T>
__device__ __forceinline__ v4f raw_to_f32(v2u v, const T*) {
  struct Pack { T h[4]; };
  const Pack pk = __builtin_bit_cast(Pack, v);
  return v4f{to_f32(pk.h[0]), to_f32(pk.h[1]), to_f32(pk.h[2]), to_f32(pk.h[3])};
}
template <typename FeatT, int B_VECS>
struct StageRegs32 {
  v4f a[2];
  typename RawVec<FeatT>::type b[B_VECS];
};

// Raw loads only (addresses clamped into the crop): the zeroing of channels >= C / positions
// >= HW happens in store_stage32, one iteration later -- a select placed here makes the compiler
// wait for the load right behind the barrier, in front of the MFMAs it is meant to hide under.
template <typename FeatT, int B_VECS, bool NHWC>
__device__ __forceinline__ void load_stage32(const StageSrc<FeatT>& s, const float* w32, int stage,
                                             StageRegs32<FeatT, B_VECS>& r) {
  const int c0 = stage * kKC;
#pragma unroll
  for (int i = 0; i < 2; ++i)  // 64 rows x 32 ch of this stage: contiguous 8 KiB
    r.a[i] = *reinterpret_cast<const v4f*>(w32 + (size_t)stage * (kRows * kKC) + (size_t)(s.tid + i * 256) * 4);
#pragma unroll
  for (int i = 0; i < B_VECS; ++i) {
    const int v = s.tid + i * 256;
    bool ok;
    size_t off;
    if constexpr (NHWC) {
      const int pos = v >> 3, c4 = v & 7;
      ok = pos < s.HW && c0 + c4 * 4 < s.C;
      off = (size_t)pos * s.C + c0 + c4 * 4;
    } else {
      const int row = v / s.vec_per_row, q = v - row * s.vec_per_row;
      ok = v < s.b_total && c0 + row < s.C;
      off = (size_t)(c0 + row) * s.HW + q * 4;
    }
    r.b[i] = load4_raw<FeatT>(s.fcrop + (ok ? off : 0));
  }
}

// The stage body has to stay ONE basic block (the MFMA / carry interleave is a scheduling-region
// property), so lanes without a valid element do not branch around their store: they aim it at a
// per-lane dump slot behind the tiles (`dump`, word offset from the buffer base).
template <int B_VECS, bool NHWC, typename FeatT>
__device__ __forceinline__ void store_stage32(const StageSrc<FeatT>& s, float* As_buf, float* Bs_buf,
                                              int dump, int stage,
                                              const StageRegs32<FeatT, B_VECS>& r) {
  const int c0 = stage * kKC;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int v = s.tid + i * 256;
    const int row = v >> 3, slot = v & 7;
    *reinterpret_cast<v4f*>(As_buf + row * kKC + ((slot ^ swz(row)) << 2)) = r.a[i];
  }
  const v4f zero = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < B_VECS; ++i) {
    const int v = s.tid + i * 256;
    if constexpr (NHWC) {
      const int pos = v >> 3, slot = v & 7;
      const int o = pos < s.HW ? pos * kKC + ((slot ^ swz(pos)) << 2) : dump;
      *reinterpret_cast<v4f*>(Bs_buf + o) =
          (c0 + slot * 4 < s.C) ? raw_to_f32(r.b[i], s.fcrop) : zero;
    } else {
      // transpose on the way in: this thread holds channel `row` of positions 4q .. 4q+3
      const int row = v / s.vec_per_row, q = v - row * s.vec_per_row;
      const bool ok = v < s.b_total;
      const v4f val = (c0 + row < s.C) ? raw_to_f32(r.b[i], s.fcrop) : zero;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int pos = q * 4 + e;
        const int o = pos * kKC + ((((row >> 2) ^ swz(pos)) << 2) | (row & 3));
        Bs_buf[ok ? o : dump + e] = val[e];
      }
    }
  }
}

template <int CT>
__host__ __device__ constexpr int hw_pad32() { return CT * 32 + 4; }

template <typename FeatT, int CT, bool ACC64, bool NHWC>
__global__ __launch_bounds__(256, 2) void head_fused32_kernel(
    const FeatT* __restrict__ feat, const float* __restrict__ packed, int B, int C, int H, int W,
    int J, int D, HeadGeom g, HeadScale hs, float* __restrict__ coords2d,
    float* __restrict__ coords3d_rel) {
  constexpr int TPW = (CT + 1) / 2;                  // column tiles per wave (upper bound)
  constexpr int HWP = hw_pad32<CT>();
  constexpr int A_STAGE = kRows * kKC;               // floats
  constexpr int B_STAGE = CT * 32 * kKC;             // floats
  constexpr int B_VECS = (CT * 32 * kKC / 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][64][32]
  float* Bs = smem + 2 * A_STAGE;   // [2][CT*32][32], then 256 x 16-byte dump slots (store_stage32)
  float* Ls = smem;                 // epilogue alias: [64][HWP]

  const int HW = H * W;
  const int chunk = 8 * g.n_groups;  // XCD-aware remap, as in the 16x16 core
  const int id = blockIdx.x;
  const int crop = (id / chunk) * 8 + (id % 8);
  const int grp = (id % chunk) / 8;
  if (crop >= B) return;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n_stages = g.c_pad / kKC;
  const FeatT* fcrop = feat + (size_t)crop * C * HW;
  const size_t n_w = (size_t)g.n_groups * g.c_pad * kRows;
  const float* w32 = packed + n_w + (size_t)g.n_groups * kRows + (size_t)grp * g.c_pad * kRows;
  const float* bgrp = packed + n_w + (size_t)grp * kRows;
  const int vec_per_row = HW / 4;
  const StageSrc<FeatT> src{nullptr, fcrop, C, HW, vec_per_row, kKC * vec_per_row, tid};

  // rows >= HW of the feature tile are never written: zero both buffers once
  for (int v = tid; v < 2 * B_STAGE; v += 256) Bs[v] = 0.0f;

  const int rt = wid & 1, ct0 = wid >> 1;
  const int fi = lane & 31, fg = lane >> 5;
  const int a_row = rt * 32 + fi;
  const int a_off = a_row * kKC + ((fg ^ swz(a_row)) << 2);  // ^ (u << 3) selects slot 2u + g
  int b_off[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int pos = (ct0 + 2 * t < CT ? ct0 + 2 * t : 0) * 32 + fi;  // (absent tile: any valid row)
    b_off[t] = pos * kKC + ((fg ^ swz(pos)) << 2);
  }

  // Register budget (256 VGPRs at 2 waves/SIMD): f64 carry accumulators take 32 per tile.
  //   2 tiles/wave, carry mode: one f32 partial per tile, 16-channel chains, the carry of a chain
  //     runs under the other tile's next chain.
  //   1 tile/wave, carry mode ("SC"): ONE carry per 32-channel stage.  The stage's two 16-channel
  //     chunks run as independent f32 chains (sub-accumulators a, b), are added in f32 (one more
  //     rounding, at the magnitude of a 32-channel sum) and that sum goes into f64: half the
  //     f64 converts/adds of carrying each 16-channel chain, which is what the carry costs
  //     (ablation: +8 us of 33 at B = 64, however evenly it is spread under the MFMAs).
  //     A single 32-channel chain would halve them too but is 3x less accurate (16 dependent
  //     roundings; 1.6e-3 vs 4.9e-4 mm on golden case s256_c1280_peaked).
  constexpr bool SC = ACC64 && TPW == 1;
  constexpr int PS = SC ? 2 : 1;
  constexpr int NCH = 2 * TPW;  // chains per stage: (chunk h, tile t), n = h * TPW + t

  using AccT = typename std::conditional<ACC64, double, float>::type;
  AccT acc[ACC64 ? TPW : 1][16];
  if constexpr (ACC64) {
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0;
  }
  // 1 tile/wave: two f32 accumulators per tile.  SC mode: sub-accumulator h = chunk h of the stage
  // (two independent 16-channel chains).  16-bit features (no carry): even / odd MFMAs of the one
  // long chain -- anything issued between two MFMAs on the SAME accumulator costs ~40 cycles
  // (MI355X_MICROARCH.md, instruction timings) and this loop puts fragment reads, global loads and
  // LDS stores exactly there; measured 34.6 -> 33.2 us at B = 64.
  constexpr int NSUB = TPW == 1 ? 2 : 1;
  f32x16 part[PS][TPW][NSUB];  // ACC64: short-chain partials; else part[0] is the accumulator
#pragma unroll
  for (int h = 0; h < PS; ++h)
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int e = 0; e < NSUB; ++e) part[h][t][e] = f32x16{0};

  auto tile_on = [&](int t) { return (CT % 2 == 0) || (ct0 + 2 * t < CT); };

  // fragments: [0..1] = channels 0..15 of the stage (chunk 0), [2..3] = channels 16..31 (chunk 1).
  // Chunk 1 is consumed one iteration late (see the loop), so it starts as zeros.
  v4f af[4], bf[TPW][4];
#pragma unroll
  for (int u = 2; u < 4; ++u) {
    af[u] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TPW; ++t) bf[t][u] = v4f{0.f, 0.f, 0.f, 0.f};
  }

#define HEAD32_READ(U0, U1)                                                                       \
  _Pragma("unroll") for (int u = (U0); u < (U1); ++u) {                                           \
    af[u] = *reinterpret_cast<const v4f*>(Ab + (a_off ^ (u << 3)));                               \
    _Pragma("unroll") for (int t = 0; t < TPW; ++t)                                               \
        bf[t][u] = *reinterpret_cast<const v4f*>(Bb + (b_off[t] ^ (u << 3)));                     \
  }
// (odd CT: the waves of the second column-tile pair run their last tile on tile 0's data and drop
//  the result -- branch-free, and the stage barrier waits for the 2-tile waves anyway)
#define HEAD32_MFMA1(P_, T_, E_, U_, S_, FIRST_)                                                  \
  part[P_][T_][E_] = __builtin_amdgcn_mfma_f32_32x32x2f32(                                        \
      af[U_][S_], bf[T_][U_][S_], (FIRST_) ? f32x16{0} : part[P_][T_][E_], 0, 0, 0);
// two elements of a finished chain into the f64 accumulators; the empty asm pins the add inside
// this basic block (otherwise it is sunk past the barrier, where no MFMA is in flight to hide it)
#define HEAD32_CARRY(P_, T_, R0, R1)                                                              \
  _Pragma("unroll") for (int r = (R0); r < (R1); ++r) {                                           \
    if constexpr (NSUB == 2)                                                                      \
      acc[T_][r] += (double)(part[P_][T_][0][r] + part[P_][T_][1][r]);                            \
    else                                                                                          \
      acc[T_][r] += (double)part[P_][T_][0][r];                                                   \
    asm volatile("" : "+v"(acc[T_][r]));                                                          \
  }
// MFMAs K0..K1-1 of chain (chunk H_, tile T_) = 16 channels of one tile.  In carry mode the
// previous chain's 16 elements are folded into f64 under the matrix pipe: nothing behind MFMA 0
// (the previous chain's last MFMA is still in flight then), 2-3 elements behind each of the other
// seven, and a scheduling fence per slot so that the even spread survives the compiler (a slot
// holding more VALU than one MFMA lasts -- 64 cycles -- idles the matrix pipe).
#define HEAD32_CHAIN(H_, T_, K0, K1)                                                              \
  {                                                                                               \
    /* (used with one partial set: 2 tiles/wave in carry mode -- the previous chain is the other  \
       tile's -- and the carry-free 16-bit mode) */                                               \
    constexpr int n_ = (H_) * TPW + (T_), pn_ = (n_ + NCH - 1) % NCH, pt_ = pn_ % TPW;            \
    _Pragma("unroll") for (int k = (K0); k < (K1); ++k) {                                         \
      HEAD32_MFMA1(0, T_, k % NSUB, 2 * (H_) + k / 4, k % 4, ACC64 && k < NSUB)                   \
      if constexpr (ACC64 && !(MTR_ABLATE & 8)) {                                                 \
        constexpr int e0_[9] = {0, 0, 2, 4, 6, 8, 11, 14, 16};                                    \
        HEAD32_CARRY(0, pt_, e0_[k], e0_[k + 1])                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                        \
      }                                                                                           \
    }                                                                                             \
  }

// ---- SC mode.  Chunk H_ (sub-accumulator H_) of the stage with parity P_, MFMAs K0..K1-1; with
// CP_ >= 0 the finished stage held in partial set CP_ is folded into f64 underneath (nothing
// behind MFMA 0: the set's last MFMA is still in flight then; 2-3 elements behind the others).
#define HEAD32_SC_CHUNK(P_, H_, K0, K1, CP_)                                                      \
  _Pragma("unroll") for (int k = (K0); k < (K1); ++k) {                                           \
    part[P_][0][H_] = __builtin_amdgcn_mfma_f32_32x32x2f32(                                       \
        af[2 * (H_) + k / 4][k % 4], bf[0][2 * (H_) + k / 4][k % 4],                              \
        k == 0 ? f32x16{0} : part[P_][0][H_], 0, 0, 0);                                           \
    if constexpr ((CP_) >= 0 && !(MTR_ABLATE & 8)) {                                              \
      constexpr int e0_[9] = {0, 0, 2, 4, 6, 8, 11, 14, 16};                                      \
      constexpr int cp_ = (CP_) >= 0 ? (CP_) : 0;                                                 \
      _Pragma("unroll") for (int r = e0_[k]; r < e0_[k + 1]; ++r) {                               \
        acc[0][r] += (double)(part[cp_][0][0][r] + part[cp_][0][1][r]);                           \
        asm volatile("" : "+v"(acc[0][r]));                                                       \
      }                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
  }
// Same iteration structure as HEAD32_ITER below; PAR = parity of the stage (a literal: the
// partial sets must be indexed statically).
// (Staggering the LDS stores by wave -- one wave at a time on the CU's store path -- measured
//  slower, 40.9 vs 38.3 us at B = 64: the ~6 us the stores cost is not queueing between waves.)
#define HEAD32_ITER_SC(S, PAR, LD, ST)                                                            \
  {                                                                                               \
    const int s_ = (S);                                                                           \
    const float* Ab = As + (PAR) * A_STAGE;                                                       \
    const float* Bb = Bs + (PAR) * B_STAGE;                                                       \
    __syncthreads();                                                                              \
    HEAD32_READ(0, 2)                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    if (!(MTR_ABLATE & (4 | 32)))                                                                 \
      load_stage32<FeatT, B_VECS, NHWC>(src, w32, min(s_ + 2, n_stages - 1), LD);                 \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    HEAD32_SC_CHUNK((PAR) ^ 1, 1, 0, 8, -1)                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    HEAD32_READ(2, 4)                                                                             \
    HEAD32_SC_CHUNK(PAR, 0, 0, 4, (PAR) ^ 1)                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    if (!(MTR_ABLATE & (4 | 16)))                                                                 \
      store_stage32<B_VECS, NHWC>(src, As + ((PAR) ^ 1) * A_STAGE, Bs + ((PAR) ^ 1) * B_STAGE,    \
                                  (1 + (PAR)) * B_STAGE + tid * 4, min(s_ + 1, n_stages - 1),     \
                                  ST);                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    HEAD32_SC_CHUNK(PAR, 0, 4, 8, (PAR) ^ 1)                                                      \
  }

  // One iteration = one 32-channel stage, ONE barrier, and the matrix pipe never drains across it:
  //   barrier                      stage s visible in LDS buffer s & 1
  //   read chunk-0 fragments of s  \  the LDS latency is covered by the chunk-1 chains of stage
  //   issue the global loads of    |  s-1, whose fragments were read before the barrier
  //     stage s+kAhead (set LD)    |
  //   chunk-1 chains of stage s-1  /
  //   read chunk-1 fragments of s
  //   chunk-0 chains of stage s, with the LDS stores of stage s+1 (register set ST, landed
  //   iterations ago) issued in the middle -- not at the end, where they would sit between the
  //   last MFMA and the barrier.
  // With one workgroup per CU (B = 64: one wave per SIMD) nothing else hides those gaps; ending
  // the stage with stores + barrier + fragment reads measured 8 us of 32 (ablation).
#define HEAD32_ITER(S, LD, ST)                                                                    \
  {                                                                                               \
    const int s_ = (S);                                                                           \
    const int buf = s_ & 1;                                                                       \
    const float* Ab = As + buf * A_STAGE;                                                         \
    const float* Bb = Bs + buf * B_STAGE;                                                         \
    __syncthreads();                                                                              \
    if (!(MTR_ABLATE & 2)) {                                                                      \
      HEAD32_READ(0, 2)                                                                           \
      __builtin_amdgcn_sched_barrier(0); /* fragment reads first, then the load addresses */      \
    }                                                                                             \
    /* (past the end: reload the last stage, never consumed -- keeps the body branch-free; a    \
       uniform branch around the load measured no better) */                                      \
    if (!(MTR_ABLATE & (4 | 32)))                                                                 \
      load_stage32<FeatT, B_VECS, NHWC>(src, w32, min(s_ + kAhead, n_stages - 1), LD);            \
    if (!(MTR_ABLATE & 2)) {                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      HEAD32_CHAIN(1, 0, 0, 8)                                                                    \
      if constexpr (TPW == 2) HEAD32_CHAIN(1, 1, 0, 8)                                            \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      HEAD32_READ(2, 4)                                                                           \
      if constexpr (TPW == 2) HEAD32_CHAIN(0, 0, 0, 8) else HEAD32_CHAIN(0, 0, 0, 4)              \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    if (!(MTR_ABLATE & (4 | 16)))                                                                 \
      store_stage32<B_VECS, NHWC>(src, As + (buf ^ 1) * A_STAGE, Bs + (buf ^ 1) * B_STAGE,        \
                                  (1 + buf) * B_STAGE + tid * 4, min(s_ + 1, n_stages - 1), ST);  \
    if (MTR_ABLATE & 16) { /* loads only: wait for them where the stores would have */           \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(ST.a[i]));              \
      _Pragma("unroll") for (int i = 0; i < B_VECS; ++i) asm volatile("" ::"v"(ST.b[i]));         \
    }                                                                                             \
    if (!(MTR_ABLATE & 2)) {                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      if constexpr (TPW == 2) HEAD32_CHAIN(0, 1, 0, 8) else HEAD32_CHAIN(0, 0, 4, 8)              \
    }                                                                                             \
  }

  // kAhead stages of global loads in flight per workgroup (register sets, rotated statically):
  // a load is issued right behind the barrier of iteration s and consumed in the middle of
  // iteration s + kAhead - 1, which has to cover ~1-2 us of latency when nothing else runs on the CU
  constexpr int kAhead = 2;  // (3 measured no better: the loop is not load-latency bound)
  StageRegs32<FeatT, B_VECS> regs[kAhead];
#pragma unroll
  for (int i = 0; i < kAhead; ++i)
    load_stage32<FeatT, B_VECS, NHWC>(src, w32, min(i, n_stages - 1), regs[i]);
  __syncthreads();  // zero fill done
  store_stage32<B_VECS, NHWC>(src, As, Bs, 2 * B_STAGE + tid * 4, 0, regs[0]);
  if constexpr (SC) {
    for (int s = 0; s < n_stages; s += 2) {
      HEAD32_ITER_SC(s, 0, regs[0], regs[1])
      if (s + 1 < n_stages) HEAD32_ITER_SC(s + 1, 1, regs[1], regs[0])
    }
    // drain: chunk 1 of the last stage, then that stage's carry (its parity is a run-time value)
    if ((n_stages - 1) & 1) {
      HEAD32_SC_CHUNK(1, 1, 0, 8, -1)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] += (double)(part[1][0][0][r] + part[1][0][1][r]);
    } else {
      HEAD32_SC_CHUNK(0, 1, 0, 8, -1)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] += (double)(part[0][0][0][r] + part[0][0][1][r]);
    }
  } else {
    for (int s = 0; s < n_stages; s += 2) {
      HEAD32_ITER(s, regs[0], regs[1])
      if (s + 1 < n_stages) HEAD32_ITER(s + 1, regs[1], regs[0])
    }
    // drain: chunk 1 of the last stage, then the last chain's carry
    HEAD32_CHAIN(1, 0, 0, 8)
    if constexpr (TPW == 2) HEAD32_CHAIN(1, 1, 0, 8)
    if constexpr (ACC64) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[TPW - 1][r] += (double)part[PS - 1][TPW - 1][0][r];
    }
  }
  __syncthreads();  // every wave is done reading the tiles: the logits may overwrite them
#undef HEAD32_ITER
#undef HEAD32_ITER_SC
#undef HEAD32_SC_CHUNK
#undef HEAD32_CHAIN
#undef HEAD32_MFMA1
#undef HEAD32_CARRY
#undef HEAD32_READ

  // ---- epilogue 1: logits (+bias) -> LDS [64][HWP].  C/D layout of f32 32x32x2: col = l & 31,
  //   row = 8 * (reg / 4) + 4 * (l >> 5) + reg % 4
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (!tile_on(t)) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rt * 32 + 8 * (r >> 2) + 4 * fg + (r & 3);
      const int col = (ct0 + 2 * t) * 32 + fi;
      if constexpr (ACC64)
        Ls[row * HWP + col] = (float)(acc[t][r] + (double)bgrp[row]);
      else
        Ls[row * HWP + col] =
            part[0][t][0][r] + (NSUB == 2 ? part[0][t][NSUB - 1][r] : 0.0f) + bgrp[row];
    }
  }
  __syncthreads();

  if (MTR_ABLATE & 1) {  // no decode: one store per workgroup keeps the GEMM alive
    if (tid == 0) coords2d[(size_t)crop * J * 2 + grp] = Ls[tid];
    return;
  }
  decode_group_from_lds<ACC64, (CT > 2 ? 4 : 2)>(Ls, HWP, grp, g, crop, J, D, H, W, hs, coords2d, coords3d_rel, wid,
                               lane);
}

// =====================================================================================
// 8-wave variant of the 32x32 core for SMALL launches (fewer workgroups than ~2 per CU; config 2's
// B = 64 gives 192 workgroups on 256 CUs).  There the 4-wave kernel leaves one wave per SIMD and
// every in-order stall of that wave (LDS store issue, carry VALU, waits) idles the matrix pipe.
// Here a workgroup has two waves per SIMD: waves 0-3 take channels 0..15 of every 32-channel
// stage, waves 4-7 channels 16..31 (same 2x2 tile assignment), so each wave issues 8 of the
// stage's 16 MFMAs per tile and the other wave's MFMAs fill its stalls; the two K-halves are
// added through LDS once, before the decode.  Same LDS tiles, same staging volume, same f32
// chains: a wave's chain is 16 channels (8 MFMAs), chains of two consecutive stages are added in
// f32 and carried once (the SC pairing, over stages instead of chunks).
template <typename FeatT, bool ACC64, bool NHWC>
__global__ __launch_bounds__(512, 2) void head_fused32w8_kernel(
    const FeatT* __restrict__ feat, const float* __restrict__ packed, int B, int C, int H, int W,
    int J, int D, HeadGeom g, HeadScale hs, float* __restrict__ coords2d,
    float* __restrict__ coords3d_rel) {
  constexpr int CT = 2;
  constexpr int HWP = hw_pad32<CT>();
  constexpr int A_STAGE = kRows * kKC;    // floats
  constexpr int B_STAGE = CT * 32 * kKC;  // floats
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                // [2][64][32]
  float* Bs = smem + 2 * A_STAGE;  // [2][64][32], then 512 x 16-byte dump slots
  float* Ls = smem;                // epilogue alias: [64][HWP]

  const int HW = H * W;
  const int chunk = 8 * g.n_groups;
  const int id = blockIdx.x;
  const int crop = (id / chunk) * 8 + (id % 8);
  const int grp = (id % chunk) / 8;
  if (crop >= B) return;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kh = wid >> 2, wt = wid & 3;  // K-half, tile
  const int n_stages = g.c_pad / kKC;
  const FeatT* fcrop = feat + (size_t)crop * C * HW;
  const size_t n_w = (size_t)g.n_groups * g.c_pad * kRows;
  const float* w32 = packed + n_w + (size_t)g.n_groups * kRows + (size_t)grp * g.c_pad * kRows;
  const float* bgrp = packed + n_w + (size_t)grp * kRows;
  const int vec_per_row = HW / 4;

  for (int v = tid; v < 2 * B_STAGE; v += 512) Bs[v] = 0.0f;

  const int rt = wt & 1, ct0 = wt >> 1;
  const int fi = lane & 31, fg = lane >> 5;
  const int a_row = rt * 32 + fi, b_pos = ct0 * 32 + fi;
  // slot 2u + g with u = 2 kh + {0, 1}
  const int a_off = a_row * kKC + (((fg ^ swz(a_row)) << 2) ^ (kh << 4));
  const int b_off = b_pos * kKC + (((fg ^ swz(b_pos)) << 2) ^ (kh << 4));

  // ---- staging: one 16-byte vector of each tile per thread and stage
  using RawT = typename RawVec<FeatT>::type;
  struct Regs { v4f a; RawT b; };
  const int b_total = kKC * vec_per_row;  // NCHW vectors per stage (<= 512 at 8x8)
  auto load_stage = [&](int stage, Regs& r) {
    const int c0 = stage * kKC;
    r.a = *reinterpret_cast<const v4f*>(w32 + (size_t)stage * (kRows * kKC) + (size_t)tid * 4);
    bool ok;
    size_t off;
    if constexpr (NHWC) {
      const int pos = tid >> 3, c4 = tid & 7;
      ok = pos < HW && c0 + c4 * 4 < C;
      off = (size_t)pos * C + c0 + c4 * 4;
    } else {
      const int row = tid / vec_per_row, q = tid - row * vec_per_row;
      ok = tid < b_total && c0 + row < C;
      off = (size_t)(c0 + row) * HW + q * 4;
    }
    r.b = load4_raw<FeatT>(fcrop + (ok ? off : 0));
  };
  auto store_stage = [&](int stage, int buf, const Regs& r) {
    const int c0 = stage * kKC;
    float* Ab = As + buf * A_STAGE;
    float* Bb = Bs + buf * B_STAGE;
    const int dump = (2 - buf) * B_STAGE + tid * 4;
    {
      const int row = tid >> 3, slot = tid & 7;
      *reinterpret_cast<v4f*>(Ab + row * kKC + ((slot ^ swz(row)) << 2)) = r.a;
    }
    const v4f zero = v4f{0.f, 0.f, 0.f, 0.f};
    if constexpr (NHWC) {
      const int pos = tid >> 3, slot = tid & 7;
      const int o = pos < HW ? pos * kKC + ((slot ^ swz(pos)) << 2) : dump;
      *reinterpret_cast<v4f*>(Bb + o) = (c0 + slot * 4 < C) ? raw_to_f32(r.b, fcrop) : zero;
    } else {
      const int row = tid / vec_per_row, q = tid - row * vec_per_row;
      const bool ok = tid < b_total;
      const v4f val = (c0 + row < C) ? raw_to_f32(r.b, fcrop) : zero;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int pos = q * 4 + e;
        const int o = pos * kKC + ((((row >> 2) ^ swz(pos)) << 2) | (row & 3));
        Bb[ok ? o : dump + e] = val[e];
      }
    }
  };

  // ---- accumulators
  double acc[ACC64 ? 16 : 1];
  if constexpr (ACC64) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0;
  }
  // ACC64: part[Q][sub]: pair Q = (stage / 2) & 1, sub = stage & 1; else part[0][k & 1] accumulate
  f32x16 part[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 2; ++e) part[q][e] = f32x16{0};
  v4f af[2], bf[2];  // [0]: channels 0..7 of this wave's chunk, [1]: channels 8..15 (used one iteration late)
  af[1] = bf[1] = v4f{0.f, 0.f, 0.f, 0.f};

#define W8_MFMA(Q_, E_, U_, S_, FIRST_)                                                            \
  part[Q_][E_] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[U_][S_], bf[U_][S_],                     \
                                                      (FIRST_) ? f32x16{0} : part[Q_][E_], 0, 0, 0);
#define W8_CARRY(Q_, R0, R1)                                                                      \
  _Pragma("unroll") for (int r = (R0); r < (R1); ++r) {                                           \
    acc[r] += (double)(part[Q_][0][r] + part[Q_][1][r]);                                          \
    asm volatile("" : "+v"(acc[r]));                                                              \
  }
// Iteration of stage S with literal phase PH = S & 3 (pair Q = PH >> 1, sub E = PH & 1):
//   barrier; read the first fragment pair of S; the SECOND half (4 MFMAs) of the previous stage's
//   chain from the fragments read before the barrier; read the second pair; issue the global
//   loads of S+2; first half of S's chain with the LDS stores of S+1 in the middle.
//   Carry mode: a pair (stages 2m, 2m+1) is complete after the deferred half at the top of
//   iteration 2m+2 and is folded into f64 under that iteration's four first-half MFMAs.
#define W8_ITER(S, PH, LD, ST)                                                                    \
  {                                                                                               \
    const int s_ = (S);                                                                           \
    constexpr int q_ = (PH) >> 1, e_ = (PH) & 1;                                                  \
    constexpr int pq_ = (((PH) + 3) & 3) >> 1, pe_ = (((PH) + 3) & 3) & 1; /* previous stage */   \
    const float* Ab = As + e_ * A_STAGE;                                                          \
    const float* Bb = Bs + e_ * B_STAGE;                                                          \
    __syncthreads();                                                                              \
    af[0] = *reinterpret_cast<const v4f*>(Ab + a_off);                                            \
    bf[0] = *reinterpret_cast<const v4f*>(Bb + b_off);                                            \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    load_stage(min(s_ + 2, n_stages - 1), LD);                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                               \
      if constexpr (ACC64) {                                                                      \
        W8_MFMA(pq_, pe_, 1, k, false)                                                            \
      } else {                                                                                    \
        W8_MFMA(0, k & 1, 1, k, false)                                                            \
      }                                                                                           \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    af[1] = *reinterpret_cast<const v4f*>(Ab + (a_off ^ 8));                                      \
    bf[1] = *reinterpret_cast<const v4f*>(Bb + (b_off ^ 8));                                      \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                               \
      if constexpr (ACC64) {                                                                      \
        W8_MFMA(q_, e_, 0, k, k == 0)                                                             \
        /* the pair finished by the deferred half above (stages S-2, S-1 when S is even) */       \
        if constexpr (e_ == 0) { W8_CARRY(q_ ^ 1, 4 * k, 4 * k + 4) __builtin_amdgcn_sched_barrier(0); } \
      } else {                                                                                    \
        W8_MFMA(0, k & 1, 0, k, false)                                                            \
      }                                                                                           \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    store_stage(min(s_ + 1, n_stages - 1), e_ ^ 1, ST);                                           \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    _Pragma("unroll") for (int k = 2; k < 4; ++k) {                                               \
      if constexpr (ACC64) {                                                                      \
        W8_MFMA(q_, e_, 0, k, false)                                                              \
        if constexpr (e_ == 0) { W8_CARRY(q_ ^ 1, 4 * k, 4 * k + 4) __builtin_amdgcn_sched_barrier(0); } \
      } else {                                                                                    \
        W8_MFMA(0, k & 1, 0, k, false)                                                            \
      }                                                                                           \
    }                                                                                             \
  }

  Regs regs0, regs1;
  load_stage(0, regs0);
  load_stage(min(1, n_stages - 1), regs1);
  __syncthreads();  // zero fill done
  store_stage(0, 0, regs0);
  for (int s = 0; s < n_stages; s += 4) {
    W8_ITER(s, 0, regs0, regs1)
    if (s + 1 < n_stages) W8_ITER(s + 1, 1, regs1, regs0)
    if (s + 2 < n_stages) W8_ITER(s + 2, 2, regs0, regs1)
    if (s + 3 < n_stages) W8_ITER(s + 3, 3, regs1, regs0)
  }
  // ---- drain: second half of the last stage's chain, then the open pair(s)
  {
    const int last = (n_stages - 1) & 3;  // phase of the last stage (run-time)
#define W8_DRAIN(PH)                                                                              \
    {                                                                                             \
      constexpr int q_ = (PH) >> 1, e_ = (PH) & 1;                                                \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                             \
        if constexpr (ACC64) { W8_MFMA(q_, e_, 1, k, false) } else { W8_MFMA(0, k & 1, 1, k, false) } \
      }                                                                                           \
      if constexpr (ACC64) {                                                                      \
        /* pair q_ holds the last one or two stages; when the last stage is even the previous    \
           pair (q_^1) was already folded in during that stage */                                 \
        if constexpr (e_ == 0) {                                                                  \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] += (double)part[q_][0][r];        \
        } else {                                                                                  \
          _Pragma("unroll") for (int r = 0; r < 16; ++r)                                          \
              acc[r] += (double)(part[q_][0][r] + part[q_][1][r]);                                \
        }                                                                                         \
      }                                                                                           \
    }
    if (last == 0) W8_DRAIN(0) else if (last == 1) W8_DRAIN(1) else if (last == 2) W8_DRAIN(2) else W8_DRAIN(3)
#undef W8_DRAIN
  }
#undef W8_ITER
#undef W8_CARRY
#undef W8_MFMA
  __syncthreads();  // every wave is done reading the tiles

  // ---- add the two K-halves through LDS (waves 4-7 publish, waves 0-3 add), logits -> Ls
  {
    using XT = typename std::conditional<ACC64, double, float>::type;
    XT* X = reinterpret_cast<XT*>(smem);  // [4 tiles][16 regs][64 lanes]
    if (kh == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        XT v;
        if constexpr (ACC64) v = acc[r]; else v = part[0][0][r] + part[0][1][r];
        X[(wt * 16 + r) * 64 + lane] = v;
      }
    }
    __syncthreads();
    XT tot[16];
    if (kh == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        XT v;
        if constexpr (ACC64) v = acc[r]; else v = part[0][0][r] + part[0][1][r];
        tot[r] = v + X[(wt * 16 + r) * 64 + lane];
      }
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + 8 * (r >> 2) + 4 * fg + (r & 3);
        const int col = ct0 * 32 + fi;
        if constexpr (ACC64)
          Ls[row * HWP + col] = (float)(tot[r] + (double)bgrp[row]);
        else
          Ls[row * HWP + col] = tot[r] + bgrp[row];
      }
    }
    __syncthreads();
  }
  if (wid < 4)
    decode_group_from_lds<ACC64, 2>(Ls, HWP, grp, g, crop, J, D, H, W, hs, coords2d, coords3d_rel, wid,
                                    lane);
}

// =====================================================================================
// 16-bit features (the autocast backbone output): f16 / bf16 MFMA, f32 accumulate.
// The reference's GPU path runs conv_final under autocast, i.e. f16 x f16 products, f32 sums and
// logits rounded to f16 ("precision classes"); here the products and sums are
// the same and the logits stay f32 on chip.  v_mfma_f32_32x32x16_{f16,bf16} has 16x the rate of
// the f32 32x32x2 core, so this kernel is a staging loop: 64-channel stages, one 16-byte LDS read
// per operand and MFMA, one barrier per stage.
//   LDS tiles are K-contiguous, [row][64 ch] = 128 B rows of eight 16-byte slots, slot ^= swz(row)
//   (same swizzle, same conflict analysis as the f32 32x32 core: a slot is one lane's operand);
//   lane (i = l & 31, g = l >> 5) of MFMA u reads slot 2u + g of row i: channels 16u + 8g .. + 7.
//   NHWC features are copied 16 B at a time.  NCHW features are transposed in registers: a thread
//   loads one dword (positions 2p, 2p + 1) from each of 8 consecutive channels -- a wave reads
//   whole 128-byte rows -- and v_perm_b32 packs the low / high halves into the two positions'
//   8-channel slots: two ds_write_b128 instead of sixteen ds_write_b16.
using v4u = __attribute__((ext_vector_type(4))) unsigned;
using h16x8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <typename T> struct Mfma16;
template <> struct Mfma16<__half> {
  static __device__ __forceinline__ f32x16 run(v4u a, v4u b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a),
                                                  __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mfma16<__hip_bfloat16> {
  static __device__ __forceinline__ f32x16 run(v4u a, v4u b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

constexpr int kKH = 64;  // channels per stage of the 16-bit core

// packed (16-bit feature dtypes) = the f32 sections above, then
//   [n_groups][ceil(C / 64)][64 rows][64 ch] weights rounded to the feature dtype
template <typename T>
__global__ void head_pack16_kernel(const float* __restrict__ w, int C, int J, int D, HeadGeom g,
                                   int n_st, T* __restrict__ w16) {
  const int per = 1 + D;
  const size_t total = (size_t)g.n_groups * n_st * kRows * kKH;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((t / ((size_t)kKH * kRows)) % n_st) * kKH + (int)(t % kKH);
    const int row = (int)((t / kKH) % kRows);
    const int grp = (int)(t / ((size_t)kKH * kRows * n_st));
    const int jl = row / per, k = row % per;
    const int j = grp * g.jg + jl;
    float v = 0.0f;
    if (jl < g.jg && j < J && c < C) v = w[(size_t)((k == 0) ? j : J + (k - 1) * J + j) * C + c];
    w16[t] = T(v);
  }
}

// GPW = joint groups (64-row blocks) per workgroup.  One group per workgroup re-stages the crop's
// feature tile once per group and reads 2 KiB of LDS per MFMA; with GPW groups a wave holds GPW
// weight fragments against each feature fragment (GPW x TPW MFMAs from GPW + TPW fragment reads)
// and the feature tile is staged once for all of them.  Small launches keep GPW = 1 (more, smaller
// workgroups to fill 256 CUs); the dispatch picks.
template <int GPW, int B_UNITS, bool NHWC>
struct StageRegs16 {
  v4u a[2 * GPW];
  // NHWC: one 16-byte slot per unit.  NCHW: 8 dwords per unit (8 channels x 2 positions).
  unsigned b[B_UNITS][NHWC ? 4 : 8];
};

template <typename FeatT>
struct StageSrc16 {
  const FeatT* fcrop;   // this crop's features
  const FeatT* w16[3];  // 16-bit weights of the workgroup's groups (absent group: a valid one)
  int C, HW, tid;
};

template <typename FeatT, int GPW, int B_UNITS, bool NHWC>
__device__ __forceinline__ void load_stage16(const StageSrc16<FeatT>& s, int stage,
                                             StageRegs16<GPW, B_UNITS, NHWC>& r) {
  const int c0 = stage * kKH;
#pragma unroll
  for (int i = 0; i < 2 * GPW; ++i)  // 64 rows x 64 ch of this stage and group: contiguous 8 KiB
    r.a[i] = *reinterpret_cast<const v4u*>(s.w16[i >> 1] + (size_t)stage * (kRows * kKH) +
                                           (size_t)(s.tid + (i & 1) * 256) * 8);
#pragma unroll
  for (int i = 0; i < B_UNITS; ++i) {
    const int v = s.tid + i * 256;
    if constexpr (NHWC) {
      const int pos = v >> 3, slot = v & 7;
      const bool ok = pos < s.HW && c0 + slot * 8 < s.C;
      const v4u x = *reinterpret_cast<const v4u*>(s.fcrop + (ok ? (size_t)pos * s.C + c0 + slot * 8 : 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) r.b[i][e] = x[e];
    } else {
      const int pairs = s.HW >> 1;
      const int kg = v / pairs, pp = v - kg * pairs;  // 8-channel group, position pair
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c0 + kg * 8 + e;
        const bool ok = kg < 8 && c < s.C;
        r.b[i][e] = *reinterpret_cast<const unsigned*>(s.fcrop + (ok ? (size_t)c * s.HW + 2 * pp : 0));
      }
    }
  }
}

// `dump`: byte offset from Bs_buf of this lane's 16-byte dump slot (lanes without a valid element
// store there instead of branching)
template <typename FeatT, int GPW, int B_UNITS, bool NHWC>
__device__ __forceinline__ void store_stage16(const StageSrc16<FeatT>& s, char* As_buf, char* Bs_buf,
                                              int dump, int stage,
                                              const StageRegs16<GPW, B_UNITS, NHWC>& r) {
  const int c0 = stage * kKH;
#pragma unroll
  for (int i = 0; i < 2 * GPW; ++i) {
    const int v = s.tid + (i & 1) * 256;
    const int row = (i >> 1) * kRows + (v >> 3), slot = v & 7;
    *reinterpret_cast<v4u*>(As_buf + row * 128 + ((slot ^ swz(row)) << 4)) = r.a[i];
  }
  const v4u zero = v4u{0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < B_UNITS; ++i) {
    const int v = s.tid + i * 256;
    if constexpr (NHWC) {
      const int pos = v >> 3, slot = v & 7;
      const int o = pos < s.HW ? pos * 128 + ((slot ^ swz(pos)) << 4) : dump;
      const v4u x = v4u{r.b[i][0], r.b[i][1], r.b[i][2], r.b[i][3]};
      *reinterpret_cast<v4u*>(Bs_buf + o) = (c0 + slot * 8 < s.C) ? x : zero;
    } else {
      const int pairs = s.HW >> 1;
      const int kg = v / pairs, pp = v - kg * pairs;
      unsigned d[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = (c0 + kg * 8 + e < s.C) ? r.b[i][e] : 0u;
      v4u lo, hi;  // position 2pp: low halves; 2pp + 1: high halves; channel e at half e
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lo[e] = __builtin_amdgcn_perm(d[2 * e + 1], d[2 * e], 0x05040100u);
        hi[e] = __builtin_amdgcn_perm(d[2 * e + 1], d[2 * e], 0x07060302u);
      }
      const int p0 = 2 * pp, p1 = 2 * pp + 1;
      const bool ok = kg < 8;
      *reinterpret_cast<v4u*>(Bs_buf + (ok ? p0 * 128 + ((kg ^ swz(p0)) << 4) : dump)) = lo;
      *reinterpret_cast<v4u*>(Bs_buf + (ok ? p1 * 128 + ((kg ^ swz(p1)) << 4) : dump)) = hi;
    }
  }
}

template <int CT, bool NHWC>
__host__ __device__ constexpr int h16_b_units() {
  // NHWC: CT*32 positions x 8 slots.  NCHW: (HW / 2 <= CT*16) pairs x 8 channel groups.
  return ((NHWC ? CT * 32 * 8 : CT * 16 * 8) + 255) / 256;
}

// developer-only build knobs (tools/experiments/ablate_head16.py); all 0 in the product
#ifndef MTR_H16_AHEAD
#define MTR_H16_AHEAD 0     // prefetch depth override
#endif
#ifndef MTR_H16_ABLATE
#define MTR_H16_ABLATE 0    // 1: no global loads in the K loop, 2: no MFMA, 4: no LDS stores, 8: no LDS reads
#endif
#ifndef MTR_H16_MINWAVES
#define MTR_H16_MINWAVES 1  // __launch_bounds__ second argument
#endif

template <typename FeatT, int CT, int GPW, bool NHWC>
__global__ __launch_bounds__(256, MTR_H16_MINWAVES) void head_fused16_kernel(
    const FeatT* __restrict__ feat, const float* __restrict__ packed, int B, int C, int H, int W,
    int J, int D, HeadGeom g, HeadScale hs, float* __restrict__ coords2d,
    float* __restrict__ coords3d_rel) {
  constexpr int TPW = (CT + 1) / 2;
  constexpr int HWP = hw_pad32<CT>();
  constexpr int A_STAGE = GPW * kRows * 128;  // bytes
  constexpr int B_STAGE = CT * 32 * 128;      // bytes
  constexpr int B_UNITS = h16_b_units<CT, NHWC>();
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* As = reinterpret_cast<char*>(smem);   // [2][GPW*64][128 B]
  char* Bs = As + 2 * A_STAGE;                // [2][CT*32][128 B], then 256 x 16-byte dump slots
  float* Ls = smem;                           // epilogue alias: [64][HWP], one group at a time

  const int HW = H * W;
  const int wg_per_crop = (g.n_groups + GPW - 1) / GPW;
  const int chunk = 8 * wg_per_crop;  // XCD-aware remap, as in the f32 cores
  const int id = blockIdx.x;
  const int crop = (id / chunk) * 8 + (id % 8);
  const int grp0 = ((id % chunk) / 8) * GPW;
  if (crop >= B) return;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n_st = (C + kKH - 1) / kKH;
  const size_t n_w = (size_t)g.n_groups * g.c_pad * kRows;
  const float* bias = packed + n_w;
  const FeatT* w16 = reinterpret_cast<const FeatT*>(packed + 2 * n_w + (size_t)g.n_groups * kRows);
  StageSrc16<FeatT> src;
  src.fcrop = feat + (size_t)crop * C * HW;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    src.w16[k] = w16 + (size_t)min(grp0 + (k < GPW ? k : 0), g.n_groups - 1) * n_st * (kRows * kKH);
  src.C = C; src.HW = HW; src.tid = tid;

  // rows >= HW of the feature tile are never written: zero both buffers once
  for (int v = tid; v < 2 * B_STAGE / 16; v += 256)
    reinterpret_cast<v4u*>(Bs)[v] = v4u{0u, 0u, 0u, 0u};

  // wave (rp, cp): row tiles rp, rp + 2, .. (one per group), column tiles cp, cp + 2, ..
  const int rp = wid & 1, cp = wid >> 1;
  const int fi = lane & 31, fg = lane >> 5;
  int a_off[GPW], b_off[TPW];  // ^ (u << 5) selects slot 2u + g
  bool on[TPW];
#pragma unroll
  for (int k = 0; k < GPW; ++k) {
    const int row = (2 * k + rp) * 32 + fi;
    a_off[k] = row * 128 + ((fg ^ swz(row)) << 4);
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    on[t] = cp + 2 * t < CT;
    const int pos = (on[t] ? cp + 2 * t : 0) * 32 + fi;
    b_off[t] = pos * 128 + ((fg ^ swz(pos)) << 4);
  }

  f32x16 acc[GPW][TPW];
#pragma unroll
  for (int k = 0; k < GPW; ++k)
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[k][t] = f32x16{0};

  // kAhead register sets of global loads in flight (rotated statically).  Measured on MI355X
  // (tools/experiments/ablate_head16.py): depth 2 - 4 is no faster than 1 at any launch
  // size, and the registers it takes cost a resident workgroup on the wide tiles (J = 122, 12x12:
  // 505 us at depth 2 vs 324 us at depth 1) -- what hides the load latency of this short loop is a
  // second workgroup on the CU, not a deeper queue in this one.
  constexpr int kAhead = MTR_H16_AHEAD ? MTR_H16_AHEAD : 1;
  const int dump = 2 * B_STAGE + tid * 16;  // byte offset from Bs of this lane's dump slot
  StageRegs16<GPW, B_UNITS, NHWC> regs[kAhead];
#pragma unroll
  for (int k = 0; k < kAhead; ++k)
    load_stage16<FeatT, GPW, B_UNITS, NHWC>(src, min(k, n_st - 1), regs[k]);
  __syncthreads();  // zero fill done
  store_stage16<FeatT, GPW, B_UNITS, NHWC>(src, As, Bs, dump, 0, regs[0]);

  // iteration s = s0 + k: regs[k] held stage s (stored during iteration s - 1) and is refilled
  // with stage s + kAhead; regs[(k + 1) % kAhead] holds stage s + 1 and is stored into the other
  // buffer, which every wave finished reading before this iteration's barrier.
  for (int s0 = 0; s0 < n_st; s0 += kAhead) {
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
      const int st = s0 + k;
      if (st >= n_st) break;
      __syncthreads();
      const int cur = st & 1, nxt = cur ^ 1;
      // unconditional (clamped) loads and stores: a load inside a branch makes the compiler merge
      // "issued" and "not issued" at the join, and the vmcnt it then puts in front of the stores
      // waits for the loads just issued -- the prefetch distance collapses to zero
      if (!(MTR_H16_ABLATE & 1))
        load_stage16<FeatT, GPW, B_UNITS, NHWC>(src, min(st + kAhead, n_st - 1), regs[k]);
      const char* Ab = As + cur * A_STAGE;
      const char* Bb = Bs + cur * B_STAGE;
#pragma unroll
      for (int uh = 0; uh < 2; ++uh) {  // two 32-channel halves: half the live fragment registers
        v4u af[GPW][2], bf[TPW][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int q = 0; q < GPW; ++q)
            af[q][u] = (MTR_H16_ABLATE & 8) ? v4u{(unsigned)a_off[q], 1u, 2u, (unsigned)st}
                : *reinterpret_cast<const v4u*>(Ab + (a_off[q] ^ ((2 * uh + u) << 5)));
#pragma unroll
          for (int t = 0; t < TPW; ++t)
            bf[t][u] = (MTR_H16_ABLATE & 8) ? v4u{(unsigned)b_off[t], 1u, 2u, (unsigned)st}
                : *reinterpret_cast<const v4u*>(Bb + (b_off[t] ^ ((2 * uh + u) << 5)));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int t = 0; t < TPW; ++t)  // (an absent tile recomputes tile 0: no branch; the
#pragma unroll                           //  other waves' tiles bound the stage anyway)
            for (int q = 0; q < GPW; ++q) {
              if (MTR_H16_ABLATE & 2) {
                acc[q][t][0] += __builtin_bit_cast(float, af[q][u][0] ^ bf[t][u][0]);
              } else {
                acc[q][t] = Mfma16<FeatT>::run(af[q][u], bf[t][u], acc[q][t]);
              }
            }
      }
      // behind the MFMA issue: the wait for the loads overlaps the matrix pipe draining
      // (after the last stage: a clamped stage into the idle buffer)
      if (!(MTR_H16_ABLATE & 4))
        store_stage16<FeatT, GPW, B_UNITS, NHWC>(src, As + nxt * A_STAGE, Bs + nxt * B_STAGE,
                                                 dump - nxt * B_STAGE, st + 1, regs[(k + 1) % kAhead]);
    }
  }

  // ---- epilogue, one group at a time (the logits of one group alias the staging tiles):
  //   logits (+bias) -> LDS [64][HWP]; this wave's row tile of group q is 2q + rp
#pragma unroll
  for (int q = 0; q < GPW; ++q) {
    __syncthreads();  // the tiles (q = 0) / the previous group's logits are no longer read
    if (grp0 + q >= g.n_groups) break;
    const float* bgrp = bias + (size_t)(grp0 + q) * kRows;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if (!on[t]) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rp * 32 + 8 * (r >> 2) + 4 * fg + (r & 3);
        Ls[row * HWP + (cp + 2 * t) * 32 + fi] = acc[q][t][r] + bgrp[row];
      }
    }
    __syncthreads();
    decode_group_from_lds<false, (CT > 2 ? 4 : 2)>(Ls, HWP, grp0 + q, g, crop, J, D, H, W, hs,
                                                   coords2d, coords3d_rel, wid, lane);
  }
}

// ---- the same kernel with the staging done by the memory pipeline: global_load_lds_dwordx4
// (gfx950) writes 16 bytes per lane straight into LDS at M0 + lane * 16 -- no staging registers,
// no ds_write pass, no VALU between the global load and the tile.  A wave-wide load fills 8 tile
// rows (1 KiB); the XOR swizzle moves to the SOURCE side: the lane that owns LDS slot s' of row r
// fetches channel slot s' ^ swz(r), which stays inside the row's 128-byte line.  NHWC features
// with C % 64 == 0 only (a partial last stage would read the next position's channels; with
// registers in between they are zeroed, here they cannot be).  Positions >= HW: those lanes are
// masked off and the rows keep their zero fill.
// One barrier per stage: the loads of stage s + 1 are issued right after the fragment reads of
// stage s and land under its MFMAs; the compiler waits for them (vmcnt) in front of the barrier.
// The builtin has to sit in a __device__ function: used directly in the kernel template (or in a
// lambda there) it compiles for the device but the host pass drops the kernel's handle, and the
// library then fails to load with an undefined symbol.
__device__ __forceinline__ void dma16_to_lds(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(src, lds_wave_base, 16, 0, 0);  // lane L -> base + 16 L
}

// two transposing 8-byte LDS reads = one 8-channel MFMA operand (semantics: see the kernel)
__device__ __forceinline__ v4u lds_read_tr16_pair(const char* p0, const char* p1) {
  using trv = __attribute__((ext_vector_type(4))) short;
  using lds_trv = __attribute__((address_space(3))) trv;
  struct Two { trv a, b; };
  return __builtin_bit_cast(v4u, Two{__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_trv*)p0),
                                     __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_trv*)p1)});
}

#define HEAD16_DMA_ISSUE(STAGE, AB, BB)                                                           \
  {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2 * GPW; ++i)                                           \
      dma16_to_lds(a_src[i] + (size_t)(STAGE) * (kRows * kKH), (AB) + (i * 4 + wid) * 1024);      \
    _Pragma("unroll") for (int i = 0; i < CT; ++i)                                                \
      if (b_on[i])                                                                                \
        dma16_to_lds(b_src[i] + (size_t)(STAGE) * b_stage_elems, (BB) + (i * 4 + wid) * 1024);    \
  }

template <typename FeatT, int CT, int GPW, bool NHWC>
__global__ __launch_bounds__(256) void head_fused16dma_kernel(
    const FeatT* __restrict__ feat, const float* __restrict__ packed, int B, int C, int H, int W,
    int J, int D, HeadGeom g, HeadScale hs, float* __restrict__ coords2d,
    float* __restrict__ coords3d_rel) {
  constexpr int TPW = (CT + 1) / 2;
  constexpr int HWP = hw_pad32<CT>();
  constexpr int A_STAGE = GPW * kRows * 128;  // bytes
  constexpr int B_STAGE = CT * 32 * 128;      // bytes
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* As = reinterpret_cast<char*>(smem);   // [2][GPW*64][128 B]
  char* Bs = As + 2 * A_STAGE;                // [2][CT*32][128 B]
  float* Ls = smem;                           // epilogue alias: [64][HWP], one group at a time

  const int HW = H * W;
  const int wg_per_crop = (g.n_groups + GPW - 1) / GPW;
  const int chunk = 8 * wg_per_crop;
  const int id = blockIdx.x;
  const int crop = (id / chunk) * 8 + (id % 8);
  const int grp0 = ((id % chunk) / 8) * GPW;
  if (crop >= B) return;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n_st = C / kKH;
  const size_t n_w = (size_t)g.n_groups * g.c_pad * kRows;
  const float* bias = packed + n_w;
  const FeatT* w16 = reinterpret_cast<const FeatT*>(packed + 2 * n_w + (size_t)g.n_groups * kRows);
  const FeatT* fcrop = feat + (size_t)crop * C * HW;

  for (int v = tid; v < 2 * B_STAGE / 16; v += 256)
    reinterpret_cast<v4u*>(Bs)[v] = v4u{0u, 0u, 0u, 0u};

  // per-lane sources of this wave's loads, stage 0.  Load i of the weights covers tile rows
  // (i * 4 + wid) * 8 .. + 7, load i of the features positions (i * 4 + wid) * 8 .. + 7; lane L is
  // (row + (L >> 3), LDS slot L & 7).
  const int lr = lane >> 3, ls = lane & 7;
  const FeatT* a_src[2 * GPW];
#pragma unroll
  for (int i = 0; i < 2 * GPW; ++i) {
    const int row = (i * 4 + wid) * 8 + lr;            // 0 .. 64 * GPW - 1
    const int grp = min(grp0 + (row >> 6), g.n_groups - 1);
    a_src[i] = w16 + (size_t)grp * n_st * (kRows * kKH) + (row & 63) * kKH + ((ls ^ swz(row)) << 3);
  }
  const FeatT* b_src[CT];
  bool b_on[CT];
  // NCHW: the tile keeps the memory layout, [channel][position] rows of HW * 2 bytes, and the
  // transpose happens in the LDS read (ds_read_b64_tr_b16).  16-byte chunk j (8 positions) of
  // channel row k sits at chunk (j + rot(k)) % n_chunks of the row, rot(k) = 4 * ((k >> 1) & 1):
  // the four channels a 16-lane group reads together then cover all 64 banks.
  const int n_chunks = HW >> 3;
  const size_t b_stage_elems = NHWC ? (size_t)kKH : (size_t)kKH * HW;
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    if constexpr (NHWC) {
      const int pos = (i * 4 + wid) * 8 + lr;
      b_on[i] = pos < HW;
      b_src[i] = fcrop + (size_t)(b_on[i] ? pos : 0) * C + ((ls ^ swz(pos)) << 3);
    } else {
      const int cid = (i * 4 + wid) * 64 + lane;  // linear 16-byte chunk of the stage in LDS
      b_on[i] = cid < kKH * n_chunks;
      const int k = b_on[i] ? cid / n_chunks : 0, jl = b_on[i] ? cid - k * n_chunks : 0;
      const int rot = ((k >> 1) & 1) << 2;
      const int j = jl >= rot ? jl - rot : jl - rot + n_chunks;  // source chunk of LDS chunk jl
      b_src[i] = fcrop + (size_t)k * HW + j * 8;
    }
  }
  const int rp = wid & 1, cp = wid >> 1;
  const int fi = lane & 31, fg = lane >> 5;
  int a_off[GPW], b_off[TPW];
  bool on[TPW];
#pragma unroll
  for (int k = 0; k < GPW; ++k) {
    const int row = (2 * k + rp) * 32 + fi;
    a_off[k] = row * 128 + ((fg ^ swz(row)) << 4);
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    on[t] = cp + 2 * t < CT;
    const int pos = (on[t] ? cp + 2 * t : 0) * 32 + fi;
    if constexpr (NHWC) {
      b_off[t] = pos * 128 + ((fg ^ swz(pos)) << 4);
    } else {
      // ds_read_b64_tr_b16 (probed, tools/experiments/tr_probe.hip): within a 16-lane group, lane
      // j receives element (j % 4) of the 8 bytes addressed by lanes (j / 4) + 4 i, i = 0..3.
      // So reader lane r = q + 4 i of group G addresses channel i (+ 4 per second read, + 8 g,
      // + 16 u) at positions P .. P + 3, P = tile + 16 (G & 1) + 4 q, and lane (n = tile + l % 32,
      // g = l / 32) ends up with channels 16 u + 8 g + 0..7 of position n: the MFMA operand.
      const int G = lane >> 4, r = lane & 15, q = r & 3, ci = r >> 2;
      const int P = (on[t] ? cp + 2 * t : 0) * 32 + 16 * (G & 1) + 4 * q;
      const int Pc = P < HW ? P : 0;  // (padding columns of the last tile: any valid data)
      int jl = (Pc >> 3) + (((ci >> 1) & 1) << 2);
      jl = jl >= n_chunks ? jl - n_chunks : jl;
      b_off[t] = (8 * fg + ci) * (HW * 2) + jl * 16 + (Pc & 7) * 2;
    }
  }
  const int tr_pitch4 = 4 * HW * 2;  // bytes between the two transposing reads of a fragment

  f32x16 acc[GPW][TPW];
#pragma unroll
  for (int k = 0; k < GPW; ++k)
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[k][t] = f32x16{0};

  __syncthreads();  // zero fill done
  HEAD16_DMA_ISSUE(0, As, Bs)
  for (int st = 0; st < n_st; ++st) {
    __syncthreads();  // stage st has landed; every wave finished reading the other buffer
    const int cur = st & 1;
    const char* Ab = As + cur * A_STAGE;
    const char* Bb = Bs + cur * B_STAGE;
    v4u af[GPW][4], bf[TPW][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int q = 0; q < GPW; ++q)
        af[q][u] = *reinterpret_cast<const v4u*>(Ab + (a_off[q] ^ (u << 5)));
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        if constexpr (NHWC) {
          bf[t][u] = *reinterpret_cast<const v4u*>(Bb + (b_off[t] ^ (u << 5)));
        } else {
          const char* p = Bb + b_off[t] + u * (4 * tr_pitch4);
          bf[t][u] = lds_read_tr16_pair(p, p + tr_pitch4);
        }
      }
    }
    // (behind the last stage: a repeat into the idle buffer)
    HEAD16_DMA_ISSUE(min(st + 1, n_st - 1), As + (cur ^ 1) * A_STAGE, Bs + (cur ^ 1) * B_STAGE)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < GPW; ++q)
          acc[q][t] = Mfma16<FeatT>::run(af[q][u], bf[t][u], acc[q][t]);
  }

#pragma unroll
  for (int q = 0; q < GPW; ++q) {
    __syncthreads();
    if (grp0 + q >= g.n_groups) break;
    const float* bgrp = bias + (size_t)(grp0 + q) * kRows;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if (!on[t]) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rp * 32 + 8 * (r >> 2) + 4 * fg + (r & 3);
        Ls[row * HWP + (cp + 2 * t) * 32 + fi] = acc[q][t][r] + bgrp[row];
      }
    }
    __syncthreads();
    decode_group_from_lds<false, (CT > 2 ? 4 : 2)>(Ls, HWP, grp0 + q, g, crop, J, D, H, W, hs,
                                                   coords2d, coords3d_rel, wid, lane);
  }
}

template <int CT, int GPW>
constexpr size_t head16_lds_bytes() {
  constexpr size_t stage = 2 * ((size_t)GPW * kRows * 128 + (size_t)CT * 32 * 128) + 256 * 16;
  constexpr size_t logits = (size_t)kRows * hw_pad32<CT>() * sizeof(float);
  return stage > logits ? stage : logits;
}

template <int CT>
constexpr size_t head32_lds_bytes() {
  constexpr size_t stage = 2 * ((size_t)kRows * kKC + (size_t)CT * 32 * kKC) + 256 * 4;  // + dump slots
  constexpr size_t logits = (size_t)kRows * hw_pad32<CT>();
  return (stage > logits ? stage : logits) * sizeof(float);
}

template <int NT, bool NHWC>
constexpr size_t head_lds_bytes() {
  constexpr size_t b_stage = NHWC ? (size_t)NT * 16 * kKP : (size_t)kKC * hw_pad(NT);
  constexpr size_t stage = 2 * ((size_t)kKC * kRowsPad + b_stage);
  constexpr size_t logits = (size_t)kRows * hw_pad(NT);
  return (stage > logits ? stage : logits) * sizeof(float);
}

template <typename FeatT, int NT, bool NHWC>
static int launch_head(const void* feat, const float* packed, int B, int C, int H, int W, int J,
                       int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d,
                       hipStream_t stream) {
  constexpr size_t lds = head_lds_bytes<NT, NHWC>();
  // f32 features -> f64 accumulate (parity with the fp32 CPU reference); 16-bit -> f32 MFMA
  auto kern = head_fused_kernel<FeatT, NT, std::is_same<FeatT, float>::value, NHWC>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int chunk = 8 * g.n_groups;
  const long long blocks = (long long)((B + 7) / 8) * chunk;
  if (blocks > 0x7fffffffLL) return MTR_E_SHAPE;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, (const FeatT*)feat,
                     packed, B, C, H, W, J, D, g, hs, c2d, c3d);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

template <typename FeatT, int CT, bool NHWC>
static int launch_head32(const void* feat, const float* packed, int B, int C, int H, int W, int J,
                         int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d,
                         hipStream_t stream) {
  constexpr size_t lds = head32_lds_bytes<CT>();
  auto kern = head_fused32_kernel<FeatT, CT, std::is_same<FeatT, float>::value, NHWC>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int chunk = 8 * g.n_groups;
  const long long blocks = (long long)((B + 7) / 8) * chunk;
  if (blocks > 0x7fffffffLL) return MTR_E_SHAPE;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, (const FeatT*)feat,
                     packed, B, C, H, W, J, D, g, hs, c2d, c3d);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

template <typename FeatT, bool NHWC>
static int launch_head32w8(const void* feat, const float* packed, int B, int C, int H, int W, int J,
                           int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d,
                           hipStream_t stream) {
  constexpr size_t lds = (2 * ((size_t)kRows * kKC + 64 * kKC) + 512 * 4) * sizeof(float);
  auto kern = head_fused32w8_kernel<FeatT, std::is_same<FeatT, float>::value, NHWC>;
  const int chunk = 8 * g.n_groups;
  const long long blocks = (long long)((B + 7) / 8) * chunk;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, stream, (const FeatT*)feat, packed,
                     B, C, H, W, J, D, g, hs, c2d, c3d);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

// MTR_HEAD_DMA=0: NHWC 16-bit features staged through registers like NCHW ones (default: staged by
// global_load_lds when C % 64 == 0; measured 5 - 15 % faster at every launch size)
static bool use_dma16() {
  static const bool v = [] {
    const char* e = getenv("MTR_HEAD_DMA");
    return !(e && e[0] == '0');
  }();
  return v;
}

template <typename FeatT, int CT, int GPW, bool NHWC>
static int launch_head16(const void* feat, const float* packed, int B, int C, int H, int W, int J,
                         int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d,
                         hipStream_t stream) {
  if constexpr (std::is_same<FeatT, float>::value) {
    return MTR_E_DTYPE;
  } else {
    constexpr size_t lds = head16_lds_bytes<CT, GPW>();
    const int chunk = 8 * ((g.n_groups + GPW - 1) / GPW);
    const long long blocks = (long long)((B + 7) / 8) * chunk;
    if (blocks > 0x7fffffffLL) return MTR_E_SHAPE;
    {
      // NHWC: any map; NCHW: whole 16-byte chunks per channel row (H*W % 8 == 0, at least the 8
      // chunks the bank rotation assumes)
      if (use_dma16() && C % kKH == 0 && (NHWC || ((H * W) % 8 == 0 && H * W >= 64))) {
        auto dma = head_fused16dma_kernel<FeatT, CT, GPW, NHWC>;
        if (lds > 64 * 1024) {
          hipError_t e = hipFuncSetAttribute((const void*)dma,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          if (e != hipSuccess) return (int)e;
        }
        MTR_CLEAR_STALE();
        hipLaunchKernelGGL(dma, dim3((unsigned)blocks), dim3(256), lds, stream, (const FeatT*)feat,
                           packed, B, C, H, W, J, D, g, hs, c2d, c3d);
        MTR_CHECK_LAUNCH();
        return MTR_OK;
      }
    }
    auto kern = head_fused16_kernel<FeatT, CT, GPW, NHWC>;
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kern,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    MTR_CLEAR_STALE();
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, (const FeatT*)feat,
                       packed, B, C, H, W, J, D, g, hs, c2d, c3d);
    MTR_CHECK_LAUNCH();
    return MTR_OK;
  }
}

// MTR_HEAD_GPW=1/2/3 forces the joint groups per workgroup of the 16-bit kernel
static int force_gpw() {
  static const int v = [] {
    const char* e = getenv("MTR_HEAD_GPW");
    return e && e[0] >= '1' && e[0] <= '3' ? e[0] - '0' : 0;
  }();
  return v;
}

template <typename FeatT, int CT, bool NHWC>
static int dispatch_head16(const void* feat, const float* packed, int B, int C, int H, int W, int J,
                           int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d,
                           hipStream_t stream) {
  // accumulators: GPW x ceil(CT / 2) tiles of 16 registers per wave
  constexpr int kMaxGpw = CT <= 2 ? 3 : (CT <= 6 ? 2 : 1);
  int gpw = 1;
  const long long crops8 = (long long)((B + 7) / 8) * 8;
  for (int cand = 2; cand <= kMaxGpw; ++cand) {
    // several groups per workgroup once the launch still fills the chip (>= 4 workgroups per CU)
    // and the groups divide without an idle remainder worse than the gain
    const int wgs = (g.n_groups + cand - 1) / cand;
    if (crops8 * wgs >= 1024 && wgs * cand - g.n_groups <= (g.n_groups >= 6 ? 1 : 0)) gpw = cand;
  }
  if (force_gpw()) gpw = force_gpw() < kMaxGpw ? force_gpw() : kMaxGpw;
  if constexpr (kMaxGpw >= 3)
    if (gpw == 3) return launch_head16<FeatT, CT, 3, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  if constexpr (kMaxGpw >= 2)
    if (gpw == 2) return launch_head16<FeatT, CT, 2, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  return launch_head16<FeatT, CT, 1, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
}

// MTR_HEAD_H16=0: 16-bit features go through the f32 cores (widened in staging, f32 weights)
static bool use_h16() {
  static const bool v = [] {
    const char* e = getenv("MTR_HEAD_H16");
    return !(e && e[0] == '0');
  }();
  return v;
}

// MTR_HEAD_W8=0 / 1 forces the 4-wave / 8-wave 32x32 kernel (default: 8 waves for small launches)
static int force_w8() {
  static const int v = [] {
    const char* e = getenv("MTR_HEAD_W8");
    return e ? (e[0] == '1' ? 1 : 0) : -1;
  }();
  return v;
}

// MTR_HEAD_CORE=16 forces the 16x16x4 core for every shape (A/B measurements, tools/microbench.py)
static bool force_core16() {
  static const bool v = [] {
    const char* e = getenv("MTR_HEAD_CORE");
    return e && e[0] == '1' && e[1] == '6';
  }();
  return v;
}

template <typename FeatT, bool NHWC>
static int dispatch_head(const void* feat, const float* packed, int B, int C, int H, int W, int J,
                         int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d,
                         hipStream_t stream) {
  const int HW = H * W;
  if (!std::is_same<FeatT, float>::value && use_h16() && !force_core16() && C % 8 == 0) {
    switch ((HW + 31) / 32) {
      case 1: return dispatch_head16<FeatT, 1, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
      case 2: return dispatch_head16<FeatT, 2, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
      case 3: return dispatch_head16<FeatT, 3, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
      case 4: return dispatch_head16<FeatT, 4, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
      case 5: return dispatch_head16<FeatT, 5, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
      case 6: return dispatch_head16<FeatT, 6, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
      default: return dispatch_head16<FeatT, 8, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    }
  }
  if (HW > 32 && HW <= 128 && !force_core16()) {
    if (HW <= 64) {
      // fewer than ~2 workgroups per CU: two waves per SIMD inside the workgroup instead
      const long long blocks = (long long)((B + 7) / 8) * 8 * g.n_groups;
      const bool small = blocks <= 512;
      if (force_w8() == 1 || (force_w8() < 0 && small))
        return launch_head32w8<FeatT, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
      return launch_head32<FeatT, 2, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    }
    if (HW <= 96) return launch_head32<FeatT, 3, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    return launch_head32<FeatT, 4, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  }
  if (HW <= 16) return launch_head<FeatT, 1, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  if (HW <= 32) return launch_head<FeatT, 2, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  if (HW <= 64) return launch_head<FeatT, 4, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  if (HW <= 144) return launch_head<FeatT, 9, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  if (HW <= 256) return launch_head<FeatT, 16, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  return MTR_E_SHAPE;
}

template <typename FeatT>
static int dispatch_head_layout(int layout, const void* feat, const float* packed, int B, int C,
                                int H, int W, int J, int D, const HeadGeom& g, const HeadScale& hs,
                                float* c2d, float* c3d, hipStream_t stream) {
  if (layout == MTR_NHWC)
    return dispatch_head<FeatT, true>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  return dispatch_head<FeatT, false>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
}

static int check_head_dims(int C, int J, int D) {
  if (C <= 0 || J <= 0 || D <= 0) return MTR_E_SHAPE;
  if (1 + D > kRows) return MTR_E_SHAPE;  // one joint must fit a 64-row workgroup tile
  return MTR_OK;
}

// bytes of the joint-group sections (64-row cores and the 16-bit kernel); 0 when 1 + D > 64
static size_t group_sections_bytes(int C, int J, int D, int feat_dtype) {
  if (check_head_dims(C, J, D)) return 0;
  const HeadGeom g = head_geom(C, J, D);
  size_t n = (2 * (size_t)g.n_groups * g.c_pad * kRows + (size_t)g.n_groups * kRows) * sizeof(float);
  // 16-bit feature dtypes: + the weights rounded to that dtype, for the f16 / bf16 MFMA kernel
  if (feat_dtype == MTR_F16 || feat_dtype == MTR_BF16)
    n += (size_t)g.n_groups * ((C + kKH - 1) / kKH) * kRows * kKH * 2;
  return n;
}

// MTR_HEAD_F32=groups: f32 features through the 64-row joint-group cores instead of the row-tile
// core (A/B measurements)
static bool force_group_cores() {
  static const bool v = [] {
    const char* e = getenv("MTR_HEAD_F32");
    return e && e[0] == 'g';
  }();
  return v;
}
static int rt_tiles_hint() {
  static const int v = [] {
    const char* e = getenv("MTR_HEAD_RTG");
    return e && e[0] >= '1' && e[0] <= '5' ? e[0] - '0' : 0;
  }();
  return v;
}

}  // namespace mtr

// host-only: the row plan of the row-tile core (which conv_final channel each packed row holds)
extern "C" int mtr_head_row_plan(int J, int D, int32_t* n_tiles, int32_t* tiles_per_atom,
                                 int32_t* row_channel, int capacity) {
  if (!n_tiles || !tiles_per_atom) return MTR_E_NULL;
  if (!mtr::rt_shape_ok(1, J, D)) return MTR_E_SHAPE;
  const mtr::RtGeom g = mtr::rt_geom(J, D);
  *n_tiles = g.n_tiles;
  *tiles_per_atom = g.a;
  if (row_channel) {
    if (capacity < g.n_tiles * 16) return MTR_E_WORKSPACE;
    for (int r = 0; r < g.n_tiles * 16; ++r) {
      const mtr::RtRow rr = mtr::rt_row(g, J, D, r);
      row_channel[r] = rr.kind == 0 ? -1 : (rr.kind == 1 ? rr.joint : J + rr.d * J + rr.joint);
    }
  }
  return MTR_OK;
}

// packed = [joint-group sections (f32 16x16 layout, bias, f32 32x32 layout, 16-bit weights)]
//          [row-tile section (f32 features only)]
extern "C" size_t mtr_head_packed_bytes(int C, int J, int D, int feat_dtype) {
  if (C <= 0 || J <= 0 || D <= 0) return 0;
  size_t n = mtr::group_sections_bytes(C, J, D, feat_dtype);
  if (feat_dtype == MTR_F32) n += mtr::rt_section_bytes(C, J, D);
  return n;
}

extern "C" int mtr_head_pack_weights(const float* weight, const float* bias, int C, int J, int D,
                                     int feat_dtype, void* packed, mtr_stream_t stream) {
  if (!weight || !bias || !packed) return MTR_E_NULL;
  if (feat_dtype != MTR_F32 && feat_dtype != MTR_F16 && feat_dtype != MTR_BF16) return MTR_E_DTYPE;
  if (mtr_head_packed_bytes(C, J, D, feat_dtype) == 0) return MTR_E_SHAPE;
  if ((uintptr_t)packed % 16) return MTR_E_ALIGN;
  const size_t group_bytes = mtr::group_sections_bytes(C, J, D, feat_dtype);
  if (feat_dtype == MTR_F32 && mtr::rt_shape_ok(C, J, D)) {
    int rc = mtr::rt_pack(weight, bias, C, J, D, (char*)packed + group_bytes, (hipStream_t)stream);
    if (rc) return rc;
  }
  if (group_bytes == 0) return MTR_OK;
  const mtr::HeadGeom g = mtr::head_geom(C, J, D);
  const size_t total = 2 * (size_t)g.n_groups * g.c_pad * mtr::kRows + (size_t)g.n_groups * mtr::kRows;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(mtr::head_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     weight, bias, C, J, D, g, (float*)packed);
  MTR_CHECK_LAUNCH();
  if (feat_dtype != MTR_F32) {
    const int n_st = (C + mtr::kKH - 1) / mtr::kKH;
    const size_t total16 = (size_t)g.n_groups * n_st * mtr::kRows * mtr::kKH;
    size_t blocks16 = (total16 + 255) / 256;
    if (blocks16 > 4096) blocks16 = 4096;
    void* w16 = (float*)packed + total;
    if (feat_dtype == MTR_F16)
      hipLaunchKernelGGL(mtr::head_pack16_kernel<__half>, dim3((unsigned)blocks16), dim3(256), 0,
                         (hipStream_t)stream, weight, C, J, D, g, n_st, (__half*)w16);
    else
      hipLaunchKernelGGL(mtr::head_pack16_kernel<__hip_bfloat16>, dim3((unsigned)blocks16), dim3(256),
                         0, (hipStream_t)stream, weight, C, J, D, g, n_st, (__hip_bfloat16*)w16);
    MTR_CHECK_LAUNCH();
  }
  return MTR_OK;
}

extern "C" int mtr_head_fused(const void* features, int feat_dtype, int layout, int B, int C, int H,
                              int W, const void* packed, int J, int D, const mtr_head_params* p,
                              float* coords2d, float* coords3d_rel, mtr_stream_t stream) {
  if (!features || !packed || !p || !coords2d || !coords3d_rel) return MTR_E_NULL;
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || J <= 0 || D <= 0) return MTR_E_SHAPE;
  if (layout != MTR_NCHW && layout != MTR_NHWC) return MTR_E_DTYPE;
  if (feat_dtype != MTR_F32 && feat_dtype != MTR_F16 && feat_dtype != MTR_BF16) return MTR_E_DTYPE;
  if ((H * W) % 4 != 0) return MTR_E_SHAPE;                  // 16-byte position vectors
  if (layout == MTR_NHWC && C % 4 != 0) return MTR_E_SHAPE;  // 16-byte channel vectors
  if (p->proc_side <= 0 || p->stride_test <= 0) return MTR_E_PARAM;
  if (((uintptr_t)features % 16) || ((uintptr_t)packed % 16)) return MTR_E_ALIGN;
  const mtr::HeadScale hs = mtr::make_head_scale(*p);
  hipStream_t s = (hipStream_t)stream;
  const size_t group_bytes = mtr::group_sections_bytes(C, J, D, feat_dtype);
  // f32 features: the row-tile core (any map size, D <= 80)
  if (feat_dtype == MTR_F32 && mtr::rt_shape_ok(C, J, D) &&
      !(mtr::force_group_cores() && group_bytes && H * W <= 256)) {
    if (B == 0) return MTR_OK;
    return mtr::rt_launch((const float*)features, layout, (const char*)packed + group_bytes, B, C, H,
                          W, J, D, hs, coords2d, coords3d_rel, mtr::rt_tiles_hint(), s);
  }
  // joint-group kernels: a joint's 1 + D rows inside one 64-row tile, maps of <= 256 positions
  if (group_bytes == 0 || H * W > 256) return MTR_E_SHAPE;  // -> 1x1-conv GEMM + mtr_softargmax_decode
  if (B == 0) return MTR_OK;
  const mtr::HeadGeom g = mtr::head_geom(C, J, D);
  const float* pk = (const float*)packed;
  switch (feat_dtype) {
    case MTR_F32: return mtr::dispatch_head_layout<float>(layout, features, pk, B, C, H, W, J, D, g, hs, coords2d, coords3d_rel, s);
    case MTR_F16: return mtr::dispatch_head_layout<__half>(layout, features, pk, B, C, H, W, J, D, g, hs, coords2d, coords3d_rel, s);
    case MTR_BF16: return mtr::dispatch_head_layout<__hip_bfloat16>(layout, features, pk, B, C, H, W, J, D, g, hs, coords2d, coords3d_rel, s);
    default: return MTR_E_DTYPE;
  }
}
