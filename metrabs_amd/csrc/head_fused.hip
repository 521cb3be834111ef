// K1 + K2-K4 fused: 1x1 heatmap projection on the matrix cores with the volumetric soft-argmax
// decode as the epilogue -- logits never reach HBM.  This file: the C-ABI entry points of the head
// (packing, dispatch) and the kernels for 16-BIT features; f32 features run the row-tile core in
// head_rt.hip.
//
// Replaces MetrabsHeads.forward as a whole (metrabs_pytorch/models/metrabs.py:75-85):
//   x = conv_final(inp)                               (LazyConv2d 1x1, models/metrabs.py:73,76)
//   split / rearrange 'b (d j) h w -> b d j h w'      (:78-79)
//   soft_argmax 3D + heatmap_to_metric, soft_argmax 2D + heatmap_to_image   (:80-83)
//
// Per crop the projection is a GEMM  logits[N_out, HW] = Wt[N_out, C] . feat[C, HW]  (N_out =
// J*(1+D) = 153, HW = 64, C = 1280 for EffNetV2-S/256).  Precision class follows the feature dtype:
//   * f32 features (the reference's CPU path): head_rt.hip -- f32-input MFMA over short chains
//     carried into f64, matrix-bound (76 FLOP/B against a ridge of 20-25);
//   * f16 / bf16 features (the autocast GPU path, where the reference itself rounds the logits to
//     f16): f16 / bf16 MFMA on features and weights of that dtype, f32 sums, f32 logits on chip,
//     i.e. strictly more accurate than the reference's f16 logits; a staging loop bounded by the
//     feature bytes.
//
// Decomposition of the 16-bit kernels
//   * weights are re-packed once (mtr_head_pack_weights) joint-major: joint j owns rows
//     [2D chan j, depth 0 .. D-1] so a JOINT GROUP is a contiguous <= 64-row block that can be
//     decoded without leaving the workgroup;
//   * one workgroup (4 waves) = (crop, 1..3 joint groups); K streams through LDS in 64-channel
//     stages;
//   * epilogue: accumulators (+bias) -> LDS [64][HWpad], then each half-wave decodes one joint
//     (softmax over its D slices, fp64 moment sums);
//   * 1-D grid with an XCD-aware remap: the joint groups of one crop run on the same XCD so the
//     crop's features are fetched from HBM once and re-read from that XCD's L2.
// (The first round's f32 joint-group kernels -- 16x16x4 and 32x32x2 cores, 64 padded rows per
//  group, up to 339 spilled VGPRs on 12x12 and 16x16 maps -- are gone: the row-tile core covers
//  every shape they did, faster, without scratch.)
#include <cstring>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"
#include "head_rt.h"
#include "head16.h"

namespace mtr {


// packed (16-bit feature dtypes) = [n_groups][64] bias (f32), then the 16-bit weights (below)
__global__ void head_pack_bias_kernel(const float* __restrict__ bias, int J, int D, HeadGeom g,
                                      float* __restrict__ packed) {
  const int per = 1 + D;
  const int total = g.n_groups * kRows;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int row = t % kRows, grp = t / kRows;
    const int jl = row / per, k = row % per;
    const int j = grp * g.jg + jl;
    float v = 0.0f;
    // reference channel order: n = j for the 2D map, J + d*J + j for depth slice d
    if (jl < g.jg && j < J) v = bias[(k == 0) ? j : J + (k - 1) * J + j];
    packed[t] = v;
  }
}


// LDS (40-90 KiB per workgroup) already caps residency at <= 4 waves per SIMD; asking for 2 lets the

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

// packed (16-bit feature dtypes) = [n_groups][64] bias (f32), then
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

// The same weights FRAGMENT-MAJOR (round 5, head_areg.hip): per (group, stage) an 8 KiB block
// [row block 0..1][16-channel step 0..3][lane 0..63][8 channels] -- lane (fi = row & 31, fg) of step u holds channels
// 16 u + 8 fg .. + 7 of row 32 rb + fi: exactly one lane's operand of v_mfma_f32_32x32x16, so that a wave-wide
// 16-byte-per-lane load is 1 KiB contiguous.
template <typename T>
__global__ void head_pack16_frag_kernel(const float* __restrict__ w, int C, int J, int D, HeadGeom g,
                                        int n_st, T* __restrict__ wfrag) {
  const int per = 1 + D;
  const size_t total = (size_t)g.n_groups * n_st * kRows * kKH;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (size_t)gridDim.x * blockDim.x) {
    const int idx = (int)(t % ((size_t)kKH * kRows));
    const size_t blk = t / ((size_t)kKH * kRows);
    const int st = (int)(blk % n_st), grp = (int)(blk / n_st);
    const int e = idx & 7, ln = (idx >> 3) & 63, ru = idx >> 9;
    const int rb = ru >> 2, u = ru & 3, fi = ln & 31, fg = ln >> 5;
    const int row = rb * 32 + fi, c = st * kKH + 16 * u + 8 * fg + e;
    const int jl = row / per, k = row % per;
    const int j = grp * g.jg + jl;
    float v = 0.0f;
    if (jl < g.jg && j < J && c < C) v = w[(size_t)((k == 0) ? j : J + (k - 1) * J + j) * C + c];
    wfrag[t] = T(v);
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
  const float* bias = packed;
  const FeatT* w16 = reinterpret_cast<const FeatT*>(packed + (size_t)g.n_groups * kRows);
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

#ifndef MTR_H16_EARLY_DEFAULT
#define MTR_H16_EARLY_DEFAULT -1  // -1: the library's rule (head16_early_copies); 0 / 1: force (ablation builds)
#endif
#ifndef MTR_H16_FRAG_PIPE
#define MTR_H16_FRAG_PIPE 0   // 1: explicit fragment double-buffering in the early-copies K loop (see there)
#endif
#ifndef MTR_H16_DMA_ABLATE
#define MTR_H16_DMA_ABLATE 0   // developer-only timing ablations of head_fused16dma_kernel (tools/experiments/
                               // ablate_head16dma.py): 1 = no decode, 2 = no logits store + no decode, 4 = no MFMA,
                               // 8 = no copies inside the K loop, 16 = no fragment reads
#endif
#define HEAD16_DMA_ISSUE(STAGE, AB, BB)                                                           \
  {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2 * GPW; ++i)                                           \
      dma16_to_lds(a_src[i] + (size_t)(STAGE) * (kRows * kKH), (AB) + (i * 4 + wid) * 1024);      \
    _Pragma("unroll") for (int i = 0; i < CT; ++i)                                                \
      if (b_on[i])                                                                                \
        dma16_to_lds(b_src[i] + (size_t)(STAGE) * b_stage_elems, (BB) + (i * 4 + wid) * 1024);    \
  }

// LD (round 4): 320 threads -- a FIFTH wave is the loader.  It issues every global_load_lds of a stage (the
// 4 x (2 GPW + CT) wave-copies the four MFMA waves used to issue between their fragment reads and their
// MFMAs, 60 - 185 issue cycles each beside MFMAs, MI355X_MICROARCH.md "LDS-DMA piece issue cost"), waits for
// them and meets the MFMA waves at the stage barrier; those run barrier, fragment reads, MFMAs.  Same
// stages, same MFMA order, same sums: the bits of the four-wave kernel.
// EARLY (round 4, the default since): the copies of stage s + 1 are issued FIRST in iteration s, right behind
// the barrier (inline asm: see dma16_to_lds_asm), and the fragments are read one 16-channel step at a time in
// front of that step's MFMAs instead of all 20 up front: the copies get the whole stage to land (they are
// what bounds the loop: L2 -> LDS at ~14 TB/s chip-wide, tools/experiments/ablate_head16dma.py), and the
// kernel drops from 228 + 96 registers (one workgroup per CU at 12x12, GPW 2: every phase of the workgroup
// serialised, the ablations' parts added up to the whole) to two workgroups per CU.  Same stages, same MFMA
// order per accumulator, same sums: the bits of the other variants.
// TIGHT (round 6, EARLY only; dma_staging 7): the feature stage holds the map's H*W positions instead of CT * 32 (12x12:
// 18 instead of 20 KiB; the reader lanes of the padding columns address a valid row) and the launch asks for exactly two
// stages -- one joint group per workgroup then needs 52 KiB and THREE workgroups share a CU.
template <typename FeatT, int CT, int GPW, bool NHWC, bool LD = false, bool EARLY = false, bool TIGHT = false>
__global__ __launch_bounds__(LD ? 320 : 256, EARLY ? 2 : 1) void head_fused16dma_kernel(
    const FeatT* __restrict__ feat, const float* __restrict__ packed, int B, int C, int H, int W,
    int J, int D, HeadGeom g, HeadScale hs, float* __restrict__ coords2d,
    float* __restrict__ coords3d_rel) {
  constexpr int TPW = (CT + 1) / 2;
  constexpr int NT = LD ? 320 : 256;
  constexpr int HWP = hw_pad32<CT>();
  constexpr int A_STAGE = GPW * kRows * 128;  // bytes
  constexpr int B_STAGE = CT * 32 * 128;      // bytes
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* As = reinterpret_cast<char*>(smem);   // [2][GPW*64][128 B]
  char* Bs = As + 2 * A_STAGE;                // [2][CT*32][128 B]
  float* Ls = smem;                           // epilogue alias: [64][HWP], one group at a time

  const int HW = H * W;
  const int b_stage = TIGHT ? HW * 128 : B_STAGE;   // bytes of a feature stage in LDS (NHWC: [HW][128 B]; NCHW: [64 ch][HW * 2 B])
  const int wg_per_crop = (g.n_groups + GPW - 1) / GPW;
  const int chunk = 8 * wg_per_crop;
  const int id = blockIdx.x;
  const int crop = (id / chunk) * 8 + (id % 8);
  const int grp0 = ((id % chunk) / 8) * GPW;
  if (crop >= B) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = LD && wid == 4;  // (wave-uniform)
  const int n_st = C / kKH;
  const float* bias = packed;
  const FeatT* w16 = reinterpret_cast<const FeatT*>(packed + (size_t)g.n_groups * kRows);
  const FeatT* fcrop = feat + (size_t)crop * C * HW;

  for (int v = tid; v < 2 * b_stage / 16; v += NT)
    reinterpret_cast<v4u*>(Bs)[v] = v4u{0u, 0u, 0u, 0u};

  if constexpr (LD) {
    if (is_loader) {
      // every wave-copy of a stage: (i, w) = load i of MFMA wave w in the four-wave kernel
      const int lr = lane >> 3, ls = lane & 7;
      const int n_chunks = HW >> 3;
      const size_t b_stage_elems = NHWC ? (size_t)kKH : (size_t)kKH * HW;
      const FeatT* a_src[2 * GPW * 4];
      const FeatT* b_src[CT * 4];
      bool b_on[CT * 4];
#pragma unroll
      for (int iw = 0; iw < 2 * GPW * 4; ++iw) {
        const int row = iw * 8 + lr;  // (i * 4 + w) * 8 + lr
        const int grp = min(grp0 + (row >> 6), g.n_groups - 1);
        a_src[iw] = w16 + (size_t)grp * n_st * (kRows * kKH) + (row & 63) * kKH + ((ls ^ swz(row)) << 3);
      }
#pragma unroll
      for (int iw = 0; iw < CT * 4; ++iw) {
        if constexpr (NHWC) {
          const int pos = iw * 8 + lr;
          b_on[iw] = pos < HW;
          b_src[iw] = fcrop + (size_t)(b_on[iw] ? pos : 0) * C + ((ls ^ swz(pos)) << 3);
        } else {
          const int cid = iw * 64 + lane;
          b_on[iw] = cid < kKH * n_chunks;
          const int k = b_on[iw] ? cid / n_chunks : 0, jl = b_on[iw] ? cid - k * n_chunks : 0;
          const int rot = nchw_chunk_rot(k, n_chunks);
          const int j = jl >= rot ? jl - rot : jl - rot + n_chunks;
          b_src[iw] = fcrop + (size_t)k * HW + j * 8;
        }
      }
      auto issue = [&](int stage, char* AB, char* BB) {
#pragma unroll
        for (int iw = 0; iw < 2 * GPW * 4; ++iw)
          dma16_to_lds(a_src[iw] + (size_t)stage * (kRows * kKH), AB + iw * 1024);
#pragma unroll
        for (int iw = 0; iw < CT * 4; ++iw)
          if (b_on[iw]) dma16_to_lds(b_src[iw] + (size_t)stage * b_stage_elems, BB + iw * 1024);
      };
      __syncthreads();  // zero fill done
      issue(0, As, Bs);
      for (int st = 0; st < n_st; ++st) {
        __syncthreads();  // (the compiler waits for this wave's copies of stage st in front of the barrier)
        const int cur = st & 1;
        if (st + 1 < n_st) issue(st + 1, As + (cur ^ 1) * A_STAGE, Bs + (cur ^ 1) * B_STAGE);
      }
      // the epilogue's barriers (two per joint group of the workgroup), nothing to do in between
#pragma unroll
      for (int q = 0; q < GPW; ++q) {
        __syncthreads();
        if (grp0 + q >= g.n_groups) break;
        __syncthreads();
      }
      return;
    }
  }

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
      const int rot = nchw_chunk_rot(k, n_chunks);
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
    const int pos_full = (on[t] ? cp + 2 * t : 0) * 32 + fi;
    const int pos = TIGHT && pos_full >= HW ? fi : pos_full;   // (TIGHT: the rows behind the map do not exist; any valid row)
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
      int jl = (Pc >> 3) + nchw_chunk_rot(ci, n_chunks);
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

  if constexpr (EARLY) {
    const unsigned As_a = lds_byte_addr(As), Bs_a = lds_byte_addr(Bs);
    auto issue_early = [&](int stage, int buf) {
#pragma unroll
      for (int i = 0; i < 2 * GPW; ++i)
        dma16_to_lds_asm(a_src[i] + (size_t)stage * (kRows * kKH), As_a + buf * A_STAGE + (i * 4 + wid) * 1024);
#pragma unroll
      for (int i = 0; i < CT; ++i)
        if (b_on[i])
          dma16_to_lds_asm(b_src[i] + (size_t)stage * b_stage_elems, Bs_a + buf * b_stage + (i * 4 + wid) * 1024);
    };
    __syncthreads();  // zero fill done
    issue_early(0, 0);
    for (int st = 0; st < n_st; ++st) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's copies of stage st have landed
      __syncthreads();  // ... everyone's; and every wave has consumed the fragments of stage st - 1
      const int cur = st & 1;
      if (st + 1 < n_st && !(MTR_H16_DMA_ABLATE & 8)) issue_early(st + 1, cur ^ 1);
      const char* Ab = As + cur * A_STAGE;
      const char* Bb = Bs + cur * b_stage;
      auto read_frags = [&](int u, v4u (&af)[GPW], v4u (&bf)[TPW]) {
#pragma unroll
        for (int q = 0; q < GPW; ++q)
          af[q] = (MTR_H16_DMA_ABLATE & 16) ? v4u{(unsigned)a_off[q], 1u, 2u, (unsigned)st}
                                            : *reinterpret_cast<const v4u*>(Ab + (a_off[q] ^ (u << 5)));
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          if (MTR_H16_DMA_ABLATE & 16) {
            bf[t] = v4u{(unsigned)b_off[t], 1u, 2u, (unsigned)st};
          } else if constexpr (NHWC) {
            bf[t] = *reinterpret_cast<const v4u*>(Bb + (b_off[t] ^ (u << 5)));
          } else {
            const char* p = Bb + b_off[t] + u * (4 * tr_pitch4);
            bf[t] = lds_read_tr16_pair(p, p + tr_pitch4);
          }
        }
      };
      // MTR_H16_FRAG_PIPE (round 6): the fragments of step u + 1 are requested before the MFMAs of step u are issued
      // (two register sets, scheduling barriers around the MFMA block) instead of leaving the order to the compiler,
      // which issues each ds_read one or two MFMAs ahead of its use and drains lgkmcnt at every step.  Same MFMAs in
      // the same order per accumulator.
      v4u af[MTR_H16_FRAG_PIPE ? 2 : 1][GPW], bf[MTR_H16_FRAG_PIPE ? 2 : 1][TPW];
      if (MTR_H16_FRAG_PIPE) read_frags(0, af[0], bf[0]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        constexpr int kSets = MTR_H16_FRAG_PIPE ? 2 : 1;
        if (!MTR_H16_FRAG_PIPE) read_frags(u, af[0], bf[0]);
        else if (u + 1 < 4) read_frags(u + 1, af[(u + 1) % kSets], bf[(u + 1) % kSets]);
        if (MTR_H16_FRAG_PIPE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
          for (int q = 0; q < GPW; ++q) {
            if (MTR_H16_DMA_ABLATE & 4) acc[q][t][0] += __builtin_bit_cast(float, af[u % kSets][q][0] ^ bf[u % kSets][t][0]);
            else acc[q][t] = Mfma16<FeatT>::run(af[u % kSets][q], bf[u % kSets][t], acc[q][t]);
          }
        if (MTR_H16_FRAG_PIPE) __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
  __syncthreads();  // zero fill done
  if constexpr (!LD) HEAD16_DMA_ISSUE(0, As, Bs)
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
        af[q][u] = (MTR_H16_DMA_ABLATE & 16) ? v4u{(unsigned)a_off[q], 1u, 2u, (unsigned)st}
                                             : *reinterpret_cast<const v4u*>(Ab + (a_off[q] ^ (u << 5)));
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        if (MTR_H16_DMA_ABLATE & 16) {
          bf[t][u] = v4u{(unsigned)b_off[t], 1u, 2u, (unsigned)st};
        } else if constexpr (NHWC) {
          bf[t][u] = *reinterpret_cast<const v4u*>(Bb + (b_off[t] ^ (u << 5)));
        } else {
          const char* p = Bb + b_off[t] + u * (4 * tr_pitch4);
          bf[t][u] = lds_read_tr16_pair(p, p + tr_pitch4);
        }
      }
    }
    // (behind the last stage: a repeat into the idle buffer)
    if constexpr (!LD) {
      if (!(MTR_H16_DMA_ABLATE & 8))
        HEAD16_DMA_ISSUE(min(st + 1, n_st - 1), As + (cur ^ 1) * A_STAGE, Bs + (cur ^ 1) * B_STAGE)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < GPW; ++q) {
          if (MTR_H16_DMA_ABLATE & 4) acc[q][t][0] += __builtin_bit_cast(float, af[q][u][0] ^ bf[t][u][0]);
          else acc[q][t] = Mfma16<FeatT>::run(af[q][u], bf[t][u], acc[q][t]);
        }
  }

  }  // (!EARLY)

#pragma unroll
  for (int q = 0; q < GPW; ++q) {
    __syncthreads();
    if (grp0 + q >= g.n_groups) break;
    const float* bgrp = bias + (size_t)(grp0 + q) * kRows;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if (!on[t] || (MTR_H16_DMA_ABLATE & 2)) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rp * 32 + 8 * (r >> 2) + 4 * fg + (r & 3);
        Ls[row * HWP + (cp + 2 * t) * 32 + fi] = acc[q][t][r] + bgrp[row];
      }
    }
    __syncthreads();
    if (MTR_H16_DMA_ABLATE & 3) {  // no decode: one store per workgroup keeps the GEMM alive
      if (tid == 0) coords2d[(size_t)crop * J * 2 + (grp0 + q)] = acc[q][0][0] + Ls[0];
      continue;
    }
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


// explicit dispatch options (mtr_head_options); zero / NULL = the library's own choice
struct HeadOpts {
  int rt_tiles = 0;          // f32: row tiles per workgroup for one-tile atoms (1..5)
  int rt_np = 0;             // f32: column blocks per workgroup tile (1..4), maps of > 64 positions
  int rt_ks = 0;             // f32: K groups per workgroup (1, 2), blocks of <= 3 tiles
  int rt_ld = 0;             // f32: loader-wave kernel (1 = never, 2 = whenever possible)
  int rt_split = 0;          // f32: column blocks over workgroups (1 = never, 2 = whenever possible)
  int groups_per_wg = 0;     // 16-bit: joint groups per workgroup (1..3)
  int dma = -1;              // 16-bit: -1 auto, 0 = stage through registers, 1 = global_load_lds
};

// the loader-wave instantiation of the DMA kernel: dma_staging 2 = always, 1 = never (the four-wave kernel),
// -1 (auto) = per the measured table below
static bool head16_loader_wave(const HeadOpts& opt, int ct, int gpw) {
  if (opt.dma == 2) return true;
  if (opt.dma == 1) return false;
  return false;  // (auto: never -- measured slower on every shape, profiles/r04c_head16_ab.jsonl: at two workgroups
                 //  per CU one wave issuing all 36 copies of a stage, one stage ahead, is the bottleneck)
}

// the early-copies instantiation (dma_staging 3).  The library's own choice (-1), from the A/B of nine shapes x
// two layouts on MI355X (profiles/r04i_head16_ab.jsonl, r04k_): NCHW features -- the transposing fragment reads
// are the longer phase there -- always (0 ... +14 %); NHWC features on the wide tiles only: six or more column
// tiles (16x16: +9 %), or three or more with at least six joint groups per crop (J = 122 on 12x12: +3 ... +8 %);
// the 8x8 tiles and J = 17 on 12x12 prefer the 228-register instantiation's deeper read-ahead (-1 ... -7 %).
static bool head16_early_copies(const HeadOpts& opt, bool nhwc, int ct, int n_groups) {
  if (opt.dma == 3) return true;
  if (opt.dma == 1 || opt.dma == 2) return false;
  if (MTR_H16_EARLY_DEFAULT >= 0) return MTR_H16_EARLY_DEFAULT != 0;  // (developer builds: ablate_head16dma.py)
  return !nhwc || ct >= 6 || (ct >= 3 && n_groups >= 6);
}

template <typename FeatT, int CT, int GPW, bool NHWC>
static int launch_head16(const void* feat, const float* packed, int B, int C, int H, int W, int J,
                         int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d,
                         const HeadOpts& opt, hipStream_t stream, bool tight = false) {
  constexpr size_t lds = head16_lds_bytes<CT, GPW>();
  const int chunk = 8 * ((g.n_groups + GPW - 1) / GPW);
  const long long blocks = (long long)((B + 7) / 8) * chunk;
  if (blocks > 0x7fffffffLL) return MTR_E_SHAPE;
  // staged by global_load_lds (measured 5 - 15 % faster at every launch size) when whole 64-channel
  // stages exist; NHWC: any map; NCHW: whole 16-byte chunks per channel row (H*W % 8 == 0, at
  // least the 8 chunks the bank rotation assumes)
  const bool dma_ok = C % kKH == 0 && (NHWC || ((H * W) % 8 == 0 && H * W >= 64));
  if (tight && dma_ok && (H * W) % 8 == 0) {   // early copies, tight feature stage, exactly two stages of LDS
    auto dma = head_fused16dma_kernel<FeatT, CT, GPW, NHWC, false, true, true>;
    const size_t stage2 = 2 * ((size_t)GPW * kRows * 128 + (size_t)H * W * 128);
    const size_t logits = (size_t)kRows * hw_pad32<CT>() * sizeof(float);
    const size_t lds_tight = stage2 > logits ? stage2 : logits;
    if (lds_tight > 64 * 1024) {
      const int rc = allow_dynamic_lds((const void*)dma, lds_tight);
      if (rc != MTR_OK) return rc;
    }
    MTR_CLEAR_STALE();
    hipLaunchKernelGGL(dma, dim3((unsigned)blocks), dim3(256), lds_tight, stream, (const FeatT*)feat, packed, B, C, H, W, J,
                       D, g, hs, c2d, c3d);
    MTR_CHECK_LAUNCH();
    return MTR_OK;
  }
  if (opt.dma != 0 && dma_ok && head16_early_copies(opt, NHWC, CT, g.n_groups)) {
    auto dma = head_fused16dma_kernel<FeatT, CT, GPW, NHWC, false, true>;
    if (lds > 64 * 1024) {
      const int rc = allow_dynamic_lds((const void*)dma, lds);
      if (rc != MTR_OK) return rc;
    }
    MTR_CLEAR_STALE();
    hipLaunchKernelGGL(dma, dim3((unsigned)blocks), dim3(256), lds, stream, (const FeatT*)feat,
                       packed, B, C, H, W, J, D, g, hs, c2d, c3d);
    MTR_CHECK_LAUNCH();
    return MTR_OK;
  }
  if (opt.dma != 0 && dma_ok && head16_loader_wave(opt, CT, GPW)) {
    auto dma = head_fused16dma_kernel<FeatT, CT, GPW, NHWC, true>;
    if (lds > 64 * 1024) {
      const int rc = allow_dynamic_lds((const void*)dma, lds);
      if (rc != MTR_OK) return rc;
    }
    MTR_CLEAR_STALE();
    hipLaunchKernelGGL(dma, dim3((unsigned)blocks), dim3(320), lds, stream, (const FeatT*)feat,
                       packed, B, C, H, W, J, D, g, hs, c2d, c3d);
    MTR_CHECK_LAUNCH();
    return MTR_OK;
  }
  if (opt.dma != 0 && dma_ok) {
    auto dma = head_fused16dma_kernel<FeatT, CT, GPW, NHWC>;
    if (lds > 64 * 1024) {
      const int rc = allow_dynamic_lds((const void*)dma, lds);
      if (rc != MTR_OK) return rc;
    }
    MTR_CLEAR_STALE();
    hipLaunchKernelGGL(dma, dim3((unsigned)blocks), dim3(256), lds, stream, (const FeatT*)feat,
                       packed, B, C, H, W, J, D, g, hs, c2d, c3d);
    MTR_CHECK_LAUNCH();
    return MTR_OK;
  }
  auto kern = head_fused16_kernel<FeatT, CT, GPW, NHWC>;
  if (lds > 64 * 1024) {
    const int rc = allow_dynamic_lds((const void*)kern, lds);
    if (rc != MTR_OK) return rc;
  }
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, (const FeatT*)feat,
                     packed, B, C, H, W, J, D, g, hs, c2d, c3d);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

// joint groups per workgroup of the 16-bit kernels (ct = column tiles of 32 positions)
static int head16_groups_per_wg(int B, int ct, const HeadGeom& g, const HeadOpts& opt) {
  // accumulators: GPW x ceil(CT / 2) tiles of 16 registers per wave
  const int max_gpw = ct <= 2 ? 3 : (ct <= 6 ? 2 : 1);
  int gpw = 1;
  const long long crops8 = (long long)((B + 7) / 8) * 8;
  for (int cand = 2; cand <= max_gpw; ++cand) {
    // several groups per workgroup once the launch still fills the chip (>= 4 workgroups per CU)
    // and the groups divide without an idle remainder worse than the gain
    const int wgs = (g.n_groups + cand - 1) / cand;
    if (crops8 * wgs >= 1024 && wgs * cand - g.n_groups <= (g.n_groups >= 6 ? 1 : 0)) gpw = cand;
    // many joint groups per crop (J = 122: 18) on a small launch: two groups per workgroup as soon as every
    // CU still gets one (B = 32: 288 workgroups, 45 -> 40 us; round 4)
    if (cand == 2 && g.n_groups >= 6 && crops8 * wgs >= 256 && wgs * cand - g.n_groups <= 1) gpw = cand;
  }
  if (opt.groups_per_wg >= 1) gpw = opt.groups_per_wg < max_gpw ? opt.groups_per_wg : max_gpw;
  return gpw;
}

// Round 6: ONE joint group per workgroup on a tight feature stage (52 KiB of LDS at 12x12: THREE workgroups per CU)
// against the rule above (two groups, two workgroups per CU).  In steady state two groups per workgroup are the more
// efficient (36.7 us per CU-round of 2 x 2 groups against 30.5 us per round of 3 x 1 at C = 1280; per workgroup alone
// on its CU: 30 / 17 us), but a launch is a whole number of resident rounds and the finer grain wastes less of the
// last one: configs[4]'s 32 crops 35.7 -> 32.1 us (NCHW; NHWC 33.4 -> 30.7), 64 crops 58.5 -> 52.6, J = 60 at 128
// crops 58.5 -> 52.4 -- and 48 crops 39.0 -> 43.7 the other way (profiles/r06l_head16_tight_crossover.jsonl).  The
// choice is a host-side model of the launch's rounds (whole rounds at the full-CU time + the trailing partial round,
// which overlaps its predecessor, at 0.85 of its own), taken only when it promises 7 % or more; same bits either way.
static double h16_round_model(long long n_wg, int slots, const double* t_by_residency) {
  const long long per_round = 256LL * slots, full = n_wg / per_round, rem = n_wg % per_round;
  double t = (double)full * t_by_residency[slots - 1];
  if (rem) t += t_by_residency[(rem + 255) / 256 - 1] * (full ? 0.85 : 1.0);
  return t;
}
static bool head16_tight_taken(const HeadOpts& opt, int B, int C, int H, int W, int layout, const HeadGeom& g, int ct,
                               int gpw_rule) {
  const bool ok = C % kKH == 0 && (H * W) % 8 == 0 && (layout == MTR_NHWC || H * W >= 64);
  if (!ok) return false;
  if (opt.dma == 7) return true;
  if (opt.dma != -1 || opt.groups_per_wg != 0) return false;
  if (ct != 5 || g.n_groups < 6) return false;   // (measured on 12x12 maps with 6 ... 18 joint groups)
  static const double t_gpw1[3] = {17.0, 22.2, 30.5}, t_gpw2[2] = {30.0, 36.7};
  const double tight = h16_round_model((long long)B * g.n_groups, 3, t_gpw1);
  const double rule = gpw_rule == 1 ? h16_round_model((long long)B * g.n_groups, 2, t_gpw1)
                                    : h16_round_model((long long)B * ((g.n_groups + 1) / 2), 2, t_gpw2);
  return tight < 0.93 * rule || (gpw_rule == 1 && tight <= rule);   // (one group either way: the same kernel, one more slot)
}

static size_t h16_frag_offset(int C, int J, int D);  // (defined with the blob's other sizes below)

// The weights-in-registers kernel (head_areg.hip): joint groups (= waves) per workgroup, 0 = not taken.
// dma_staging 4 forces it (groups_per_workgroup 2 ... 4, default 3).  The library's own choice (-1), from
// profiles/r05m_areg_frag.jsonl (MI355X, J = 122 on 12x12, f16, NCHW / NHWC; all variants bit-identical): four
// groups per workgroup on launches of >= 512 crops with >= 8 joint groups per crop -- 1024 crops 639 / 623 us
// against the early-copies kernel's 687 / 673; it is behind at 256 crops (194 vs 177 - 187) and below.
static int head16_areg_groups(const HeadOpts& opt, int B, int C, int H, int W, int layout, const HeadGeom& g) {
  if (!head16_areg_supported(C, H, W, layout)) return 0;
  if (opt.dma == 4) return opt.groups_per_wg >= 2 ? opt.groups_per_wg : 3;
  if (opt.dma != -1 || opt.groups_per_wg != 0) return 0;
  return (B >= 512 && g.n_groups >= 8 && (H * W + 31) / 32 == 5) ? 4 : 0;
}

// The resident-weights kernel (head_res.hip).  dma_staging 5 forces it.
static bool head16_res_taken(const HeadOpts& opt, int B, int C, int H, int W, int layout, const HeadGeom& g) {
  (void)B; (void)g;
  if (!head16_res_supported(C, H, W, layout)) return false;
  return opt.dma == 5;
}

// The two-halves kernel (head_pp.hip).  dma_staging 6 forces it.
static bool head16_pp_taken(const HeadOpts& opt, int B, int C, int H, int W, int layout, const HeadGeom& g) {
  (void)B; (void)g;
  if (!head16_pp_supported(C, H, W, layout)) return false;
  return opt.dma == 6;
}

template <typename FeatT, int CT, bool NHWC>
static int dispatch_head16(const void* feat, const float* packed, int B, int C, int H, int W, int J,
                           int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d,
                           const HeadOpts& opt, hipStream_t stream) {
  constexpr int kMaxGpw = CT <= 2 ? 3 : (CT <= 6 ? 2 : 1);
  if (head16_pp_taken(opt, B, C, H, W, NHWC ? MTR_NHWC : MTR_NCHW, g))   // eight waves, two alternating halves
    return head16_pp_launch(std::is_same<FeatT, __half>::value ? MTR_F16 : MTR_BF16, NHWC ? MTR_NHWC : MTR_NCHW, feat,
                            packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  if (head16_res_taken(opt, B, C, H, W, NHWC ? MTR_NHWC : MTR_NCHW, g))  // weights resident, persistent workgroups
    return head16_res_launch(std::is_same<FeatT, __half>::value ? MTR_F16 : MTR_BF16, NHWC ? MTR_NHWC : MTR_NCHW,
                             feat, packed, (const char*)packed + h16_frag_offset(C, J, D), B, C, H, W, J, D, g, hs,
                             c2d, c3d, stream);
  if (const int ag = head16_areg_groups(opt, B, C, H, W, NHWC ? MTR_NHWC : MTR_NCHW, g))  // weights in registers
    return head16_areg_launch(std::is_same<FeatT, __half>::value ? MTR_F16 : MTR_BF16, NHWC ? MTR_NHWC : MTR_NCHW, ag,
                              feat, packed, (const char*)packed + h16_frag_offset(C, J, D), B, C, H, W, J, D, g, hs,
                              c2d, c3d, stream);
  const int gpw = head16_groups_per_wg(B, CT, g, opt);
  const bool tight = head16_tight_taken(opt, B, C, H, W, NHWC ? MTR_NHWC : MTR_NCHW, g, CT, gpw);
  if (tight) {   // one group per workgroup (the library's own choice), or two on request (dma_staging 7)
    if constexpr (kMaxGpw >= 2)
      if (opt.dma == 7 && opt.groups_per_wg == 2)
        return launch_head16<FeatT, CT, 2, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream, true);
    return launch_head16<FeatT, CT, 1, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream, true);
  }
  if constexpr (kMaxGpw >= 3)
    if (gpw == 3) return launch_head16<FeatT, CT, 3, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream);
  if constexpr (kMaxGpw >= 2)
    if (gpw == 2) return launch_head16<FeatT, CT, 2, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream);
  return launch_head16<FeatT, CT, 1, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream);
}

template <typename FeatT, bool NHWC>
static int dispatch_head(const void* feat, const float* packed, int B, int C, int H, int W, int J,
                         int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d,
                         const HeadOpts& opt, hipStream_t stream) {
  switch ((H * W + 31) / 32) {
    case 1: return dispatch_head16<FeatT, 1, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream);
    case 2: return dispatch_head16<FeatT, 2, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream);
    case 3: return dispatch_head16<FeatT, 3, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream);
    case 4: return dispatch_head16<FeatT, 4, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream);
    case 5: return dispatch_head16<FeatT, 5, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream);
    case 6: return dispatch_head16<FeatT, 6, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream);
    default: return dispatch_head16<FeatT, 8, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream);
  }
}

template <typename FeatT>
static int dispatch_head_layout(int layout, const void* feat, const float* packed, int B, int C,
                                int H, int W, int J, int D, const HeadGeom& g, const HeadScale& hs,
                                float* c2d, float* c3d, const HeadOpts& opt, hipStream_t stream) {
  if (layout == MTR_NHWC)
    return dispatch_head<FeatT, true>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream);
  return dispatch_head<FeatT, false>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, opt, stream);
}

// 16-bit kernels: one joint's 1 + D rows inside a 64-row group, whole 16-byte (8-channel) operands
static bool h16_shape_ok(int C, int J, int D) {
  return C > 0 && J > 0 && D > 0 && 1 + D <= kRows && C % 8 == 0;
}

// bytes of the 16-bit blob: bias [n_groups][64] f32 + weights [n_groups][ceil(C/64)][64][64] 16-bit
static size_t h16_blob_bytes(int C, int J, int D) {
  if (!h16_shape_ok(C, J, D)) return 0;
  const HeadGeom g = head_geom(J, D);
  return (size_t)g.n_groups * kRows * sizeof(float) +
         (size_t)g.n_groups * ((C + kKH - 1) / kKH) * kRows * kKH * 2;
}

static size_t h16_blob_padded(int C, int J, int D) { return (h16_blob_bytes(C, J, D) + 15) & ~(size_t)15; }

// the fragment-major copy of the joint-group weights (head_areg.hip): whole 64-channel stages only
static size_t h16_frag_bytes(int C, int J, int D) {
  if (!h16_shape_ok(C, J, D) || C % kKH != 0) return 0;
  return (size_t)head_geom(J, D).n_groups * (C / kKH) * kRows * kKH * 2;
}
static size_t rt16_section_padded(int C, int J, int D) { return (rt16_section_bytes(C, J, D) + 15) & ~(size_t)15; }
static size_t h16_frag_offset(int C, int J, int D) { return h16_blob_padded(C, J, D) + rt16_section_padded(C, J, D); }

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

// f32 features: the row-tile blob (head_rt.h); f16 / bf16 features: bias + 16-bit joint-group tiles.
// 0 = this (C, J, D, dtype) has no fused kernel -> 1x1-conv GEMM + mtr_softargmax_decode.
extern "C" size_t mtr_head_packed_bytes(int C, int J, int D, int feat_dtype) {
  if (C <= 0 || J <= 0 || D <= 0) return 0;
  if (feat_dtype == MTR_F32) return mtr::rt_section_bytes(C, J, D);
  // 16-bit: [joint-group blob (1 + D <= 64, C % 8 == 0), padded to 16 bytes][row-tile section (C % 64 == 0)]
  // [the joint-group weights fragment-major (1 + D <= 64, C % 64 == 0; round 5)]
  if (feat_dtype == MTR_F16 || feat_dtype == MTR_BF16) {
    const size_t frag = mtr::h16_frag_bytes(C, J, D);
    return frag ? mtr::h16_frag_offset(C, J, D) + frag : mtr::h16_blob_padded(C, J, D) + mtr::rt16_section_bytes(C, J, D);
  }
  return 0;
}

extern "C" int mtr_head_pack_weights(const float* weight, const float* bias, int C, int J, int D,
                                     int feat_dtype, void* packed, mtr_stream_t stream) {
  if (!weight || !bias || !packed) return MTR_E_NULL;
  if (feat_dtype != MTR_F32 && feat_dtype != MTR_F16 && feat_dtype != MTR_BF16) return MTR_E_DTYPE;
  if (mtr_head_packed_bytes(C, J, D, feat_dtype) == 0) return MTR_E_SHAPE;
  if ((uintptr_t)packed % 16) return MTR_E_ALIGN;
  if (feat_dtype == MTR_F32) return mtr::rt_pack(weight, bias, C, J, D, packed, (hipStream_t)stream);
  if (mtr::rt16_section_bytes(C, J, D)) {
    const int rc = mtr::rt16_pack(weight, bias, C, J, D, feat_dtype, (char*)packed + mtr::h16_blob_padded(C, J, D),
                                  (hipStream_t)stream);
    if (rc != MTR_OK) return rc;
  }
  if (mtr::h16_blob_bytes(C, J, D) == 0) return MTR_OK;
  const mtr::HeadGeom g = mtr::head_geom(J, D);
  const int n_bias = g.n_groups * mtr::kRows;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(mtr::head_pack_bias_kernel, dim3((unsigned)((n_bias + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, bias, J, D, g, (float*)packed);
  MTR_CHECK_LAUNCH();
  const int n_st = (C + mtr::kKH - 1) / mtr::kKH;
  const size_t total16 = (size_t)g.n_groups * n_st * mtr::kRows * mtr::kKH;
  size_t blocks16 = (total16 + 255) / 256;
  if (blocks16 > 4096) blocks16 = 4096;
  void* w16 = (float*)packed + n_bias;
  if (feat_dtype == MTR_F16)
    hipLaunchKernelGGL(mtr::head_pack16_kernel<__half>, dim3((unsigned)blocks16), dim3(256), 0,
                       (hipStream_t)stream, weight, C, J, D, g, n_st, (__half*)w16);
  else
    hipLaunchKernelGGL(mtr::head_pack16_kernel<__hip_bfloat16>, dim3((unsigned)blocks16), dim3(256),
                       0, (hipStream_t)stream, weight, C, J, D, g, n_st, (__hip_bfloat16*)w16);
  MTR_CHECK_LAUNCH();
  if (mtr::h16_frag_bytes(C, J, D)) {
    void* wfrag = (char*)packed + mtr::h16_frag_offset(C, J, D);
    if (feat_dtype == MTR_F16)
      hipLaunchKernelGGL(mtr::head_pack16_frag_kernel<__half>, dim3((unsigned)blocks16), dim3(256), 0,
                         (hipStream_t)stream, weight, C, J, D, g, n_st, (__half*)wfrag);
    else
      hipLaunchKernelGGL(mtr::head_pack16_frag_kernel<__hip_bfloat16>, dim3((unsigned)blocks16), dim3(256), 0,
                         (hipStream_t)stream, weight, C, J, D, g, n_st, (__hip_bfloat16*)wfrag);
    MTR_CHECK_LAUNCH();
  }
  return MTR_OK;
}

// 16-bit features: do the joint-group kernels take the shape, or the row-tile core?
static bool head16_takes_rt(int C, int J, int D, int H, int W) {
  return !(mtr::h16_shape_ok(C, J, D) && H * W <= 256) && mtr::rt16_section_bytes(C, J, D) != 0;
}

static int parse_head_options(const mtr_head_options* caller, mtr::HeadOpts& opt) {
  if (!caller) return MTR_OK;
  // versioned by its first member: the fields inside the caller's struct_size, the library's own choice
  // (0; dma_staging: -1) for the ones behind it
  const uint32_t size = caller->struct_size;
  if (size < 8 || size > 256 || size % 4 != 0) return MTR_E_PARAM;
  mtr_head_options mine;
  memset(&mine, 0, sizeof mine);
  mine.dma_staging = -1;
  memcpy(&mine, caller, size < sizeof mine ? size : sizeof mine);
  const mtr_head_options* options = &mine;
  if (options->rt_tiles_per_workgroup < 0 || options->rt_tiles_per_workgroup > 5 ||
      options->groups_per_workgroup < 0 || options->groups_per_workgroup > 4 ||
      options->dma_staging < -1 || options->dma_staging > 7 ||
      options->rt_column_blocks < 0 || options->rt_column_blocks > 4 ||
      options->rt_k_groups < 0 || options->rt_k_groups > 2 ||
      options->rt_loader < 0 || options->rt_loader > 2 ||
      options->rt_split_column_blocks < 0 || options->rt_split_column_blocks > 2)
    return MTR_E_PARAM;
  opt.rt_tiles = options->rt_tiles_per_workgroup;
  opt.rt_np = options->rt_column_blocks;
  opt.rt_ks = options->rt_k_groups;
  opt.rt_ld = options->rt_loader;
  opt.rt_split = options->rt_split_column_blocks;
  opt.groups_per_wg = options->groups_per_workgroup;
  opt.dma = options->dma_staging;
  return MTR_OK;
}

extern "C" int mtr_head_plan(int feat_dtype, int layout, int B, int C, int H, int W, int J, int D,
                             const mtr_head_options* options, int have_workspace, mtr_head_plan_info* plan) {
  if (!plan) return MTR_E_NULL;
  *plan = mtr_head_plan_info{0, 0, 1, 0, 0};
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || J <= 0 || D <= 0 || (H * W) % 4 != 0) return MTR_E_SHAPE;
  if (layout != MTR_NCHW && layout != MTR_NHWC) return MTR_E_DTYPE;
  if (layout == MTR_NHWC && C % 4 != 0) return MTR_E_SHAPE;
  mtr::HeadOpts opt;
  const int rc = parse_head_options(options, opt);
  if (rc != MTR_OK) return rc;
  if (feat_dtype == MTR_F32) {
    if (!mtr::rt_shape_ok(C, J, D)) return MTR_E_SHAPE;
    const mtr::RtDispatch d = mtr::rt_dispatch(B, C, H, W, J, D, opt.rt_tiles, opt.rt_np, opt.rt_ks, opt.rt_ld,
                                               opt.rt_split, have_workspace != 0 && (H * W + 63) / 64 >= 2);
    *plan = mtr_head_plan_info{d.kernel, d.rtg, d.np, d.split, d.n_wg, d.model_us};
    return MTR_OK;
  }
  if (feat_dtype != MTR_F16 && feat_dtype != MTR_BF16) return MTR_E_DTYPE;
  if (head16_takes_rt(C, J, D, H, W)) {
    const mtr::RtDispatch d = mtr::rt16_dispatch(B, H, W, J, D, opt.rt_tiles, opt.rt_split,
                                                 have_workspace != 0 && (H * W + 63) / 64 >= 2);
    *plan = mtr_head_plan_info{MTR_HEAD_KERNEL_16_RT, d.rtg, 1, d.split, d.n_wg, 0.0};
    return (layout == MTR_NCHW && !have_workspace) ? MTR_E_WORKSPACE : MTR_OK;
  }
  if (!mtr::h16_shape_ok(C, J, D) || H * W > 256) return MTR_E_SHAPE;
  const mtr::HeadGeom g = mtr::head_geom(J, D);
  int ct = (H * W + 31) / 32;
  if (ct == 7) ct = 8;
  if (mtr::head16_pp_taken(opt, B, C, H, W, layout, g)) {
    plan->kernel = MTR_HEAD_KERNEL_16_PP;
    plan->tiles_per_workgroup = mtr::kPpGroupsPerWorkgroup;
    plan->workgroups = (long long)((B + 7) / 8) * 8 * ((g.n_groups + mtr::kPpGroupsPerWorkgroup - 1) / mtr::kPpGroupsPerWorkgroup);
    return MTR_OK;
  }
  if (mtr::head16_res_taken(opt, B, C, H, W, layout, g)) {
    plan->kernel = MTR_HEAD_KERNEL_16_RES;
    plan->tiles_per_workgroup = 2;
    plan->workgroups = (long long)((g.n_groups + 1) / 2) * (B < 256 / ((g.n_groups + 1) / 2) ? B : 256 / ((g.n_groups + 1) / 2));
    return MTR_OK;
  }
  if (const int ag = mtr::head16_areg_groups(opt, B, C, H, W, layout, g)) {
    plan->kernel = MTR_HEAD_KERNEL_16_AREG;
    plan->tiles_per_workgroup = ag;
    plan->workgroups = (long long)((B + 7) / 8) * 8 * ((g.n_groups + ag - 1) / ag);
    return MTR_OK;
  }
  const int gpw = mtr::head16_groups_per_wg(B, ct, g, opt);
  if (mtr::head16_tight_taken(opt, B, C, H, W, layout, g, ct, gpw)) {
    const int tg = (opt.dma == 7 && opt.groups_per_wg == 2 && ct <= 6) ? 2 : 1;
    plan->kernel = MTR_HEAD_KERNEL_16_DMA_EARLY_TIGHT;
    plan->tiles_per_workgroup = tg;
    plan->workgroups = (long long)((B + 7) / 8) * 8 * ((g.n_groups + tg - 1) / tg);
    return MTR_OK;
  }
  const bool dma_ok = C % mtr::kKH == 0 && (layout == MTR_NHWC || ((H * W) % 8 == 0 && H * W >= 64));
  plan->kernel = !(opt.dma != 0 && dma_ok) ? MTR_HEAD_KERNEL_16
                 : mtr::head16_early_copies(opt, layout == MTR_NHWC, ct, g.n_groups) ? MTR_HEAD_KERNEL_16_DMA_EARLY
                 : mtr::head16_loader_wave(opt, ct, gpw) ? MTR_HEAD_KERNEL_16_DMA_LOADER
                                                         : MTR_HEAD_KERNEL_16_DMA;
  plan->tiles_per_workgroup = gpw;
  plan->workgroups = (long long)((B + 7) / 8) * 8 * ((g.n_groups + gpw - 1) / gpw);
  return MTR_OK;
}

extern "C" size_t mtr_head_workspace_bytes(int feat_dtype, int layout, int B, int C, int H, int W, int J, int D) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || J <= 0 || D <= 0) return 0;
  if (feat_dtype == MTR_F32) return mtr::rt_workspace_bytes(B, J, D, H, W);
  if ((feat_dtype == MTR_F16 || feat_dtype == MTR_BF16) && head16_takes_rt(C, J, D, H, W))
    return mtr::rt16_workspace_bytes(B, C, J, D, H, W, layout);
  return 0;
}

extern "C" int mtr_head_fused_opts(const void* features, int feat_dtype, int layout, int B, int C,
                                   int H, int W, const void* packed, int J, int D,
                                   const mtr_head_params* p, const mtr_head_options* options,
                                   float* coords2d, float* coords3d_rel, mtr_stream_t stream) {
  // (no workspace to offer: a shape that needs one has no fused kernel HERE -> MTR_E_SHAPE, the code callers
  //  fall back to the library GEMM on)
  const int rc = mtr_head_fused_ws(features, feat_dtype, layout, B, C, H, W, packed, J, D, p, options, nullptr, 0,
                                   coords2d, coords3d_rel, stream);
  return rc == MTR_E_WORKSPACE ? MTR_E_SHAPE : rc;
}

extern "C" int mtr_head_fused_ws(const void* features, int feat_dtype, int layout, int B, int C, int H,
                                 int W, const void* packed, int J, int D, const mtr_head_params* p,
                                 const mtr_head_options* options, void* workspace, size_t workspace_bytes,
                                 float* coords2d, float* coords3d_rel, mtr_stream_t stream) {
  if (!features || !packed || !p || !coords2d || !coords3d_rel) return MTR_E_NULL;
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || J <= 0 || D <= 0) return MTR_E_SHAPE;
  if (layout != MTR_NCHW && layout != MTR_NHWC) return MTR_E_DTYPE;
  if (feat_dtype != MTR_F32 && feat_dtype != MTR_F16 && feat_dtype != MTR_BF16) return MTR_E_DTYPE;
  if ((H * W) % 4 != 0) return MTR_E_SHAPE;                  // 16-byte position vectors
  if (layout == MTR_NHWC && C % 4 != 0) return MTR_E_SHAPE;  // 16-byte channel vectors
  if (p->proc_side <= 0 || p->stride_test <= 0) return MTR_E_PARAM;
  if (((uintptr_t)features % 16) || ((uintptr_t)packed % 16)) return MTR_E_ALIGN;
  mtr::HeadOpts opt;
  {
    const int rc = parse_head_options(options, opt);
    if (rc != MTR_OK) return rc;
  }
  const mtr::HeadScale hs = mtr::make_head_scale(*p);
  hipStream_t s = (hipStream_t)stream;
  if (feat_dtype == MTR_F32) {  // the row-tile core: any map size, D <= 80
    if (!mtr::rt_shape_ok(C, J, D)) return MTR_E_SHAPE;  // -> 1x1-conv GEMM + mtr_softargmax_decode
    if (B == 0) return MTR_OK;
    if (workspace && ((uintptr_t)workspace % 8)) return MTR_E_ALIGN;
    return mtr::rt_launch((const float*)features, layout, packed, B, C, H, W, J, D, hs, coords2d,
                          coords3d_rel, opt.rt_tiles, opt.rt_np, opt.rt_ks, opt.rt_ld, opt.rt_split,
                          workspace, workspace_bytes, s);
  }
  // 16-bit: the row-tile core beyond the joint-group kernels' limits (1 + D > 64 rows, > 256 positions)
  if (head16_takes_rt(C, J, D, H, W)) {
    if (B == 0) return MTR_OK;
    if (workspace && ((uintptr_t)workspace % 16)) return MTR_E_ALIGN;
    return mtr::rt16_launch(features, feat_dtype, layout, (const char*)packed + mtr::h16_blob_padded(C, J, D), B, C, H,
                            W, J, D, hs, coords2d, coords3d_rel, opt.rt_tiles, opt.rt_split, workspace,
                            workspace_bytes, s);
  }
  // 16-bit joint-group kernels: a joint's 1 + D rows inside one 64-row tile, maps of <= 256 positions
  if (!mtr::h16_shape_ok(C, J, D) || H * W > 256) return MTR_E_SHAPE;
  if (B == 0) return MTR_OK;
  const mtr::HeadGeom g = mtr::head_geom(J, D);
  const float* pk = (const float*)packed;
  if (feat_dtype == MTR_F16)
    return mtr::dispatch_head_layout<__half>(layout, features, pk, B, C, H, W, J, D, g, hs, coords2d,
                                             coords3d_rel, opt, s);
  return mtr::dispatch_head_layout<__hip_bfloat16>(layout, features, pk, B, C, H, W, J, D, g, hs,
                                                   coords2d, coords3d_rel, opt, s);
}

extern "C" int mtr_head_fused(const void* features, int feat_dtype, int layout, int B, int C, int H,
                              int W, const void* packed, int J, int D, const mtr_head_params* p,
                              float* coords2d, float* coords3d_rel, mtr_stream_t stream) {
  return mtr_head_fused_opts(features, feat_dtype, layout, B, C, H, W, packed, J, D, p, nullptr,
                             coords2d, coords3d_rel, stream);
}
