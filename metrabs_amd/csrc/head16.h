// Shared pieces of the 16-bit joint-group head kernels (head_fused.hip, head_areg.hip): group geometry,
// the decode epilogue over logits in LDS, the MFMA wrappers and the LDS-DMA / transposing-read helpers.
#pragma once
#include "common.h"

namespace mtr {

constexpr int kRows = 64;       // rows (output channels) per joint group = 4 waves x 16

struct HeadGeom {
  int n_groups;      // joint groups
  int jg;            // joints per group (last group may hold fewer)
};

__host__ __device__ inline HeadGeom head_geom(int J, int D) {
  HeadGeom g;
  const int per = 1 + D;
  const int jg_max = kRows / per;  // >= 1 is checked by the caller
  g.n_groups = (J + jg_max - 1) / jg_max;
  g.jg = (J + g.n_groups - 1) / g.n_groups;  // balanced groups
  return g;
}

using v4f = __attribute__((ext_vector_type(4))) float;
using v2u = __attribute__((ext_vector_type(2))) unsigned;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// 16-byte slot swizzle of the K-contiguous LDS tiles: slot ^= swz(row) (found by exhaustive search:
// conflict-free for the ds_read_b128 lane groups of the 32x32 MFMA operands)
__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 1); }

template <int CT>
__host__ __device__ constexpr int hw_pad32() { return CT * 32 + 4; }

// ---- decode epilogue shared by the 16-bit kernels: logits of one joint group in LDS [64][HWP]

// (row = jl*(1+D) + {0: 2D map, 1+d: depth slice d}); a half-wave (32 lanes) per joint (<= 8 joints
// in flight).  The logits are on chip and the epilogue is a few % of the GEMM, so the f64-accumulate
// mode also takes exp in f64: the decode error then is the f32 rounding of the outputs only, which
// matters because reconstruct_absolute amplifies coords3d_rel errors ~7x.
// PV = positions per lane and step: 4 for maps of more than 64 positions, 2 below (an 8x8 map
// then keeps all 32 lanes of the half-wave busy instead of 16).
// DC: the number of depth slices when known at compile time (8: every shipped model), 0 = read from D.  Same
// operations in the same order per accumulator -- the same bits; with DC the slice loops are straight-line code:
// no loop counters, no int -> double conversions of the slice index, the eight reads and exponentials of a
// position in flight together.
#ifndef MTR_DECODE_DC8
#define MTR_DECODE_DC8 1
#endif
template <bool ACC64, int PV, int NW = 4, int DC = 0>
__device__ __forceinline__ void decode_group_from_lds_pv(const float* Ls, int HWP, int grp,
                                                         const HeadGeom& g, int crop, int J, int D,
                                                         int H, int W, const HeadScale& hs,
                                                         float* __restrict__ coords2d,
                                                         float* __restrict__ coords3d_rel, int wid,
                                                         int lane) {
  using vecf = __attribute__((ext_vector_type(PV))) float;
  const int HW = H * W;
  if (DC > 0) D = DC;
  const int per = 1 + D;
  const int li = lane & 31;
  const float rcp_w = __frcp_rn((float)W);
  for (int jl = wid * 2 + (lane >> 5); jl < g.jg; jl += 2 * NW) {   // (NW waves = 2 NW half-waves)
    const int j = grp * g.jg + jl;
    if (j >= J) continue;
    const float* row2d = Ls + (size_t)(jl * per) * HWP;
    const float* row3d = row2d + HWP;
    float m2 = -INFINITY, m3 = -INFINITY;
    for (int p = li * PV; p < HW; p += 32 * PV) {
      const vecf v = *reinterpret_cast<const vecf*>(row2d + p);
#pragma unroll
      for (int q = 0; q < PV; ++q) m2 = fmaxf(m2, v[q]);
#pragma unroll
      for (int d = 0; d < (DC > 0 ? DC : D); ++d) {
        const vecf u = *reinterpret_cast<const vecf*>(row3d + (size_t)d * HWP + p);
#pragma unroll
        for (int q = 0; q < PV; ++q) m3 = fmaxf(m3, u[q]);
      }
    }
    m2 = group_max<32>(m2);
    m3 = group_max<32>(m3);
    // exp(x - m) as ONE v_exp_f32 of fma(x, log2 e, -m log2 e) (round 4; was libm's expf: ~12 instructions
    // per logit of a VALU-bound epilogue) -- what the f32 row-tile kernel and the stand-alone decode do
    // (common.h: exp_shifted; the sums stay f64)
    const float nm2 = -m2 * kLog2e, nm3 = -m3 * kLog2e;
    double s2 = 0, sx2 = 0, sy2 = 0, s3 = 0, sx3 = 0, sy3 = 0, sz3 = 0, sz3b = 0;
    for (int p = li * PV; p < HW; p += 32 * PV) {
      const vecf v2 = *reinterpret_cast<const vecf*>(row2d + p);
      // two depth slices per pass, each with its own f64 chains (even / odd slices): the epilogue runs at
      // two waves per SIMD, where one dependent chain of f64 adds per lane is latency, not throughput
      double col[PV], colb[PV];
#pragma unroll
      for (int q = 0; q < PV; ++q) col[q] = colb[q] = 0;
      int d = 0;
#pragma unroll
      for (; d + 1 < (DC > 0 ? DC : D); d += 2) {
        const vecf ua = *reinterpret_cast<const vecf*>(row3d + (size_t)d * HWP + p);
        const vecf ub = *reinterpret_cast<const vecf*>(row3d + (size_t)(d + 1) * HWP + p);
#pragma unroll
        for (int q = 0; q < PV; ++q) {
          const double ea = ACC64 ? exp_neg64((double)ua[q] - (double)m3) : (double)exp_shifted(ua[q], nm3);
          const double eb = ACC64 ? exp_neg64((double)ub[q] - (double)m3) : (double)exp_shifted(ub[q], nm3);
          col[q] += ea;
          colb[q] += eb;
          sz3 += ea * (double)d;
          sz3b += eb * (double)(d + 1);
        }
      }
      if (d < (DC > 0 ? DC : D)) {
        const vecf ua = *reinterpret_cast<const vecf*>(row3d + (size_t)d * HWP + p);
#pragma unroll
        for (int q = 0; q < PV; ++q) {
          const double ea = ACC64 ? exp_neg64((double)ua[q] - (double)m3) : (double)exp_shifted(ua[q], nm3);
          col[q] += ea;
          sz3 += ea * (double)d;
        }
      }
#pragma unroll
      for (int q = 0; q < PV; ++q) {
        // (narrow maps wrap more than once; no integer division: exact for positions < 2^16, head_rt.hip)
        const int h = HW <= 65536 ? (int)(((float)(p + q) + 0.5f) * rcp_w) : (p + q) / W, w = (p + q) - h * W;
        const double e2 = ACC64 ? exp_neg64((double)v2[q] - (double)m2) : (double)exp_shifted(v2[q], nm2);
        const double c = col[q] + colb[q];
        s2 += e2; sx2 += e2 * w; sy2 += e2 * h;
        s3 += c; sx3 += c * w; sy3 += c * h;
      }
    }
    sz3 += sz3b;
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

// Positions per lane and step of the half-wave that decodes a joint: the choice that leaves the fewest idle
// lane-slots, ceil(HW / (32 PV)) * PV minimal (ties: the wider read).  Round 4: a 12x12 map (144 positions)
// ran PV = 4, i.e. two steps of 128 positions with 112 of the second step's 128 slots idle -- 44 % of the
// epilogue's lane-slots; at PV = 1 it is five steps of 32 with 16 idle.  The epilogue was 44 - 52 % of
// the 16-bit kernels' time (tools/experiments/head_fixed_vs_stage.py: the same launch at C = 64 ... 1280).
// UNROLL8: take the straight-line form when D == 8 (head_areg.hip, 244 registers at five column tiles, keeps the
// loops: the unrolled epilogue spilled there)
template <bool ACC64, int PVMAX, int NW = 4, bool UNROLL8 = true>
__device__ __forceinline__ void decode_group_from_lds(const float* Ls, int HWP, int grp,
                                                      const HeadGeom& g, int crop, int J, int D,
                                                      int H, int W, const HeadScale& hs,
                                                      float* __restrict__ coords2d,
                                                      float* __restrict__ coords3d_rel, int wid,
                                                      int lane) {
  const int HW = H * W;
  const int c1 = (HW + 31) / 32, c2 = (HW + 63) / 64 * 2, c4 = (HW + 127) / 128 * 4;  // lane-slots per row
  const bool d8 = MTR_DECODE_DC8 && UNROLL8 && D == 8;   // (uniform)
#define MTR_DECODE_CALL(PV_)                                                                                        \
  do {                                                                                                              \
    if (d8) decode_group_from_lds_pv<ACC64, PV_, NW, 8>(Ls, HWP, grp, g, crop, J, D, H, W, hs, coords2d,            \
                                                        coords3d_rel, wid, lane);                                   \
    else decode_group_from_lds_pv<ACC64, PV_, NW, 0>(Ls, HWP, grp, g, crop, J, D, H, W, hs, coords2d, coords3d_rel, \
                                                     wid, lane);                                                    \
  } while (0)
  if (PVMAX >= 4 && c4 <= c2 && c4 <= c1) MTR_DECODE_CALL(4);
  else if (c2 <= c1) MTR_DECODE_CALL(2);
  else MTR_DECODE_CALL(1);
#undef MTR_DECODE_CALL
}

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

__device__ __forceinline__ void dma16_to_lds(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(src, lds_wave_base, 16, 0, 0);  // lane L -> base + 16 L
}

// The same copy issued from inline asm: the compiler does not know that LDS is written, so it neither
// waits for vmcnt(0) in front of every later ds_read (what it does behind the builtin: prefetch distance
// zero for anything issued before the reads) nor orders anything for us -- the kernel waits for its own
// copies (s_waitcnt vmcnt) in front of the stage barrier.  LDS address = M0 + 16 * lane.
__device__ __forceinline__ void dma16_to_lds_asm(const void* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :
               : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(src)
               : "memory", "m0");  // (M0 is overwritten: the register allocator must know)
}
__device__ __forceinline__ unsigned lds_byte_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// NCHW feature tiles keep the memory layout [channel k][HW positions] (rows of n_chunks = HW / 8 sixteen-byte chunks)
// and are transposed by ds_read_b64_tr_b16.  Chunk j of channel row k sits at chunk (j + nchw_chunk_rot(k)) % n_chunks
// of the row, so that the FOUR channel rows a 32-lane group of the transposing read touches (k & 3 = 0 .. 3, sixteen
// dwords each) fall on four different quarters of the 64 banks.  Rows of 18 chunks (12x12 maps; round 6) start 8 banks
// apart by themselves: a rotation of 2 (k & 3) chunks spreads them 16 apart -- 0, 16, 32, 48.  The first rule,
// 4 ((k >> 1) & 1), was laid out for 8x8 maps (rows 32 banks apart: 0, 32, 16, 48) and left rows 0 / 1 and 2 / 3 of a
// 12x12 tile overlapping in 8 banks each: SQ_LDS_BANK_CONFLICT 8.85 M cycles of the kernel's 30.2 M LDS cycles at
// 256 crops of configs[4] (profiles/r06g_lds_dma3_nchw.md; NHWC tiles: 0).  Only k & 3 matters to a reader lane.
__device__ __forceinline__ int nchw_chunk_rot(int k, int n_chunks) {
  return n_chunks == 18 ? (k & 3) << 1 : ((k >> 1) & 1) << 2;
}

// two transposing 8-byte LDS reads = one 8-channel MFMA operand (semantics: see the kernel)
__device__ __forceinline__ v4u lds_read_tr16_pair(const char* p0, const char* p1) {
  using trv = __attribute__((ext_vector_type(4))) short;
  using lds_trv = __attribute__((address_space(3))) trv;
  struct Two { trv a, b; };
  return __builtin_bit_cast(v4u, Two{__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_trv*)p0),
                                     __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_trv*)p1)});
}

// head_areg.hip: the weights-in-registers variant of the joint-group kernels (a wave per joint group, gpw =
// 2 ... 4 groups per workgroup, 3 - 5 column tiles, whole 64-channel stages; reads the fragment-major section).
bool head16_areg_supported(int C, int H, int W, int layout);
int head16_areg_launch(int feat_dtype, int layout, int gpw, const void* feat, const float* bias, const void* wfrag,
                       int B, int C, int H, int W, int J, int D, const HeadGeom& g, const HeadScale& hs, float* c2d,
                       float* c3d, hipStream_t stream);

// head_res.hip: weights RESIDENT in registers, persistent workgroups (a pair of joint groups per workgroup for
// the whole launch, crops streamed through a ring of feature stages); C = 1280, 3 - 5 column tiles.
bool head16_res_supported(int C, int H, int W, int layout);
int head16_res_launch(int feat_dtype, int layout, const void* feat, const float* bias, const void* wfrag, int B, int C,
                      int H, int W, int J, int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d,
                      hipStream_t stream);

// head_pp.hip (round 6): eight waves in two alternating halves -- one half multiplies a stage while the other issues
// the next stage's copies; four joint groups per workgroup; C % 64 == 0, 3 - 5 column tiles.  Reads the row-major blob.
constexpr int kPpGroupsPerWorkgroup = 4;
bool head16_pp_supported(int C, int H, int W, int layout);
int head16_pp_launch(int feat_dtype, int layout, const void* feat, const float* packed, int B, int C, int H, int W,
                     int J, int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d, hipStream_t stream);

}  // namespace mtr
