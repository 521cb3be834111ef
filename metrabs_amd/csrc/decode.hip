// K2-K4: volumetric soft-argmax decode of heatmap logits that are already in memory.
//
// Replaces MetrabsHeads.forward after the 1x1 conv (metrabs_pytorch/models/metrabs.py:78-85):
// channel split, 'b (d j) h w -> b d j h w', joint softmax over (D,H,W) (ptu.py:47-51), the three
// marginal expectations (ptu.py:58-75), the 2D soft-argmax over (H,W) and the pixel / metric
// scaling (models/util.py:6-33) -- one pass over the logits instead of the reference's ~8.
//
// Work decomposition (HBM-bound; algorithmic bytes = J*(1+D)*H*W*sizeof(logit) + 20*J per crop):
//   * a group of LPJ lanes owns one (crop, joint); a wave owns 64/LPJ CONSECUTIVE joints, because
//     channels j, j+1, .. of one depth slice are adjacent in memory ('(d j)' ordering): at 8x8
//     (HW=64, LPJ=16, float4 per lane) every wave-wide load is one contiguous, 256-B aligned 1 KiB;
//   * an ONLINE softmax (group-wide running max m, rescale on growth) so any D (8 .. 72 ..) and any
//     H*W stream through a fixed register budget; D slices are fetched 8 at a time so 8-9
//     independent 16-B loads per lane are in flight;
//   * VALU budget (the first version was VALU-bound at 2.3k instructions per wave): exp(x-m) is one
//     fma + one v_exp_f32; the depth sums of a column run in fp32 (<= 8 terms) and are promoted to
//     fp64 once per column; the four moment sums (total, x, y, z) are fp64 and are merged across
//     the group with DPP butterflies.
#include "common.h"

namespace mtr {

// f16 logits, 8 per lane: the 16 loaded bytes stay packed.  The running maximum is taken on the
// packed halves (v_pk_max_f16: two elements per instruction, no conversion) and the exponent's
// argument is formed straight from the half (fmaf((float)h, log2e, -m log2e) = one v_fma_mix_f32),
// so an element costs fma-mix + v_exp_f32 + two accumulates instead of two conversions (the
// compiler re-converted after the group-wide maximum rather than hold 72 floats) + max + fma + ...
using h2 = __attribute__((ext_vector_type(2))) _Float16;
struct RawH8 { h2 p[4]; };

template <int AUX>
__device__ __forceinline__ RawH8 buffer_load_h8(buffer_rsrc_t rsrc, int voff_bytes, int soff_bytes) {
  const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_bytes, soff_bytes, AUX);
  static_assert(sizeof(raw) == 16, "b128 load");
  return __builtin_bit_cast(RawH8, raw);
}
// (inline asm: the builtin maximum canonicalises every operand first -- one more v_pk_max_f16 per
//  loaded dword; logits are finite, a NaN would poison the softmax either way)
__device__ __forceinline__ h2 pk_max(h2 a, h2 b) {
  h2 r;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ h2 raw_max(const RawH8& r) {
  return pk_max(pk_max(r.p[0], r.p[1]), pk_max(r.p[2], r.p[3]));
}

template <int VEC>
__device__ __forceinline__ float vec_max(const float (&v)[VEC]) {
  float r = v[0];
#pragma unroll
  for (int i = 1; i < VEC; ++i) r = fmaxf(r, v[i]);
  return r;
}

// SPAN (round 6): a wave's 64 / LPJ joint groups are CONSECUTIVE (crop, joint) pairs of the whole batch instead of
// joints of one crop -- at J = 17 the per-crop mapping left the last wave of every crop 1 / 8 (16-bit logits: eight
// joints per wave, 17 = 8 + 8 + 1) or 1 / 4 (f32: 4 + 4 + 4 + 4 + 1) occupied: 71 % / 85 % of the lanes worked.  One
// descriptor over the whole tensor (< 4 GiB; the host falls back otherwise), the crop in the per-lane offset; a
// joint's arithmetic does not change: the same bits.
template <typename T, int VEC, int LPJ, int AUX, bool SPAN>
__global__ __launch_bounds__(256) void decode_nchw_kernel(
    const T* __restrict__ logits, int B, int J, int D, int H, int W, HeadScale hs, AxisInv ai,
    float* __restrict__ coords2d, float* __restrict__ coords3d_rel) {
  constexpr int JPW = kWave / LPJ;  // joints per wave
  constexpr int CH = 8;             // depth slices fetched per round
  constexpr int kLoadAux = AUX;  // 2 = non-temporal: the logits are read exactly once
  constexpr bool kH8 = std::is_same<T, __half>::value && VEC == 8;  // packed-half path (above)
  const int HW = H * W;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(
      (int)(blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave)));
  const int crop_elems = J * (1 + D) * HW;
  int b, j;
  bool joint_ok;
  buffer_rsrc_t rsrc;
  int lane_base;  // element offset of the lane's crop inside the descriptor
  if constexpr (SPAN) {
    const long long total = (long long)B * J;
    if ((long long)wave * JPW >= total) return;  // wave-uniform
    const long long g_raw = (long long)wave * JPW + lane / LPJ;
    joint_ok = g_raw < total;
    // lanes past the last pair recompute it (valid addresses, no exec masking) and skip the final store
    const int gj = (int)(joint_ok ? g_raw : total - 1);
    b = gj / J;
    j = gj - b * J;
    rsrc = make_rsrc(uniform_ptr(logits), (unsigned)((size_t)B * crop_elems * sizeof(T)));
    lane_base = b * crop_elems;
  } else {
    const int groups_per_crop = (J + JPW - 1) / JPW;
    b = wave / groups_per_crop;
    if (b >= B) return;  // wave-uniform
    const int j_raw = (wave % groups_per_crop) * JPW + lane / LPJ;
    joint_ok = j_raw < J;
    // lanes past the last joint recompute joint J-1 (valid addresses, no exec masking) and skip
    // the final store
    j = joint_ok ? j_raw : J - 1;
    // one descriptor per wave over this crop's channels; per-lane part of the address in voff,
    // the depth-slice part is a scalar offset
    rsrc = make_rsrc(uniform_ptr(logits + (size_t)b * crop_elems), (unsigned)crop_elems * sizeof(T));
    lane_base = 0;
  }
  const int li = lane % LPJ;
  const int slice_bytes = J * HW * (int)sizeof(T);  // depth slice d sits d*J channels further

  // Running maxima are GROUP-wide (shared by the LPJ lanes of the joint), so every lane rescales
  // by the same factor and the final merge is a plain sum.
  float m3 = -INFINITY, m2 = -INFINITY;
  double s3 = 0.0, sx3 = 0.0, sy3 = 0.0, sz3 = 0.0;  // sum e, sum e*x, sum e*y, sum e*z
  double s2 = 0.0, sx2 = 0.0, sy2 = 0.0;

  const int n_iter = (HW + VEC * LPJ - 1) / (VEC * LPJ);
  for (int it = 0; it < n_iter; ++it) {
    const int p_raw = (it * LPJ + li) * VEC;
    const bool pos_ok = p_raw < HW;     // lanes past the map re-read position 0 with weight 0
    const int p0 = pos_ok ? p_raw : 0;
    // (h, w) of the VEC elements; the VEC=4 / 8 instantiations require W % VEC == 0: no wrap there
    const int h0 = p0 / W, w0 = p0 - h0 * W;
    float fx[VEC], fy[VEC];  // small integers, exact in f32; widened at use (saves 8 VGPRs)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      int w = w0 + v, h = h0;
      if (w >= W) { w -= W; ++h; }
      fx[v] = (float)w;
      fy[v] = (float)h;
    }
    const int voff = (int)((unsigned)(lane_base + j * HW + p0) * (unsigned)sizeof(T));

    // ---- the 2D heatmap load is issued together with the first round of depth slices
    float v2[VEC];
    RawH8 r2;
    if constexpr (kH8) r2 = buffer_load_h8<kLoadAux>(rsrc, voff, 0);
    else buffer_load_vec<T, VEC, kLoadAux>(rsrc, voff, 0, v2);

    for (int d0 = 0; d0 < D; d0 += CH) {
      float v3[kH8 ? 1 : CH][VEC];
      RawH8 r3[kH8 ? CH : 1];
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int d = (d0 + k < D) ? d0 + k : D - 1;  // tail rounds re-read slice D-1, weight 0
        const int soff = (J * HW) * (int)sizeof(T) + d * slice_bytes;
        if constexpr (kH8) r3[k] = buffer_load_h8<kLoadAux>(rsrc, voff, soff);
        else buffer_load_vec<T, VEC, kLoadAux>(rsrc, voff, soff, v3[k]);
      }
      // element v of the 2D row / of depth slice k as the argument of exp_shifted
      auto x2 = [&](int v) -> float {
        if constexpr (kH8) return (float)r2.p[v >> 1][v & 1]; else return v2[v];
      };
      auto x3 = [&](int k, int v) -> float {
        if constexpr (kH8) return (float)r3[k].p[v >> 1][v & 1]; else return v3[k][v];
      };
      if (d0 == 0) {
        float lm;
        if constexpr (kH8) {
          const h2 m = raw_max(r2);
          lm = fmaxf((float)m[0], (float)m[1]);
        } else {
          lm = vec_max<VEC>(v2);
        }
        const float cm = group_max<LPJ>(pos_ok ? lm : -INFINITY);
        if (cm > m2) {  // group-uniform
          const double r = (double)exp_shifted(m2, -cm * kLog2e);
          s2 *= r; sx2 *= r; sy2 *= r;
          m2 = cm;
        }
        const float nm = pos_ok ? -m2 * kLog2e : -INFINITY;  // exp2(-inf) = 0: masked lanes
        if constexpr (VEC > 1) {
          // the lane's VEC positions are consecutive in ONE map row (W % VEC == 0): x = w0 + v,
          // y = h0 -> sum e and sum v e per lane, the row / column offsets applied once
          double S = 0.0, Sv = 0.0;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const double e = (double)exp_shifted(x2(v), nm);
            S += e;
            if (v) Sv = fma(e, (double)v, Sv);
          }
          s2 += S;
          sx2 += fma(S, (double)fx[0], Sv);
          sy2 = fma(S, (double)fy[0], sy2);
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const double e = (double)exp_shifted(x2(v), nm);
            s2 += e;
            sx2 += e * (double)fx[v];
            sy2 += e * (double)fy[v];
          }
        }
      }
      {
        float cm;
        if constexpr (kH8) {
          h2 m = raw_max(r3[0]);
#pragma unroll
          for (int k = 1; k < CH; ++k) m = pk_max(m, raw_max(r3[k]));
          cm = fmaxf((float)m[0], (float)m[1]);
        } else {
          cm = vec_max<VEC>(v3[0]);
#pragma unroll
          for (int k = 1; k < CH; ++k) cm = fmaxf(cm, vec_max<VEC>(v3[k]));
        }
        cm = group_max<LPJ>(pos_ok ? cm : -INFINITY);
        if (cm > m3) {
          const double r = (double)exp_shifted(m3, -cm * kLog2e);
          s3 *= r; sx3 *= r; sy3 *= r; sz3 *= r;
          m3 = cm;
        }
        const float nm = pos_ok ? -m3 * kLog2e : -INFINITY;
        // per (h,w) column: fp32 sums over the <= 8 depth terms of this round (2 VALU per
        // element), promoted to fp64 once per column; x / y weights applied once per column
        float col[VEC], colz[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) col[v] = colz[v] = 0.0f;
        if constexpr (VEC % 2 == 0) {
          // (round 5) the two accumulations of a column PAIR are one v_pk_add_f32 and one v_pk_fma_f32 (full rate
          // on gfx950: two f32 lanes per instruction; the same IEEE operations, so the same bits) -- the 16-bit
          // instantiation is VALU-bound and a quarter of its slots were these adds
          using v2f = __attribute__((ext_vector_type(2))) float;
          v2f c2[VEC / 2], z2[VEC / 2];
#pragma unroll
          for (int v = 0; v < VEC / 2; ++v) c2[v] = z2[v] = v2f{0.0f, 0.0f};
#pragma unroll
          for (int k = 0; k < CH; ++k) {
            const float fz = (float)(d0 + k);
            const float nmk = (d0 + k < D) ? nm : -INFINITY;  // wave-uniform select
#pragma unroll
            for (int v = 0; v < VEC / 2; ++v) {
              const v2f e = v2f{exp_shifted(x3(k, 2 * v), nmk), exp_shifted(x3(k, 2 * v + 1), nmk)};
              c2[v] += e;
              z2[v] = __builtin_elementwise_fma(e, v2f{fz, fz}, z2[v]);
            }
          }
#pragma unroll
          for (int v = 0; v < VEC; ++v) { col[v] = c2[v >> 1][v & 1]; colz[v] = z2[v >> 1][v & 1]; }
        } else {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const float fz = (float)(d0 + k);
          const float nmk = (d0 + k < D) ? nm : -INFINITY;  // wave-uniform select
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float e = exp_shifted(x3(k, v), nmk);
            col[v] += e;
            colz[v] = fmaf(e, fz, colz[v]);
          }
        }
        }
        if constexpr (VEC > 1) {
          double S = 0.0, Sv = 0.0;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const double c = (double)col[v];
            S += c;
            if (v) Sv = fma(c, (double)v, Sv);
          }
          double Z = 0.0;
#pragma unroll
          for (int v = 0; v < VEC; ++v) Z += (double)colz[v];
          s3 += S;
          sx3 += fma(S, (double)fx[0], Sv);
          sy3 = fma(S, (double)fy[0], sy3);
          sz3 += Z;
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const double c = (double)col[v];
            s3 += c;
            sx3 += c * (double)fx[v];
            sy3 += c * (double)fy[v];
            sz3 += (double)colz[v];
          }
        }
      }
    }
  }

  // ---- merge the LPJ lanes of the group (same running max everywhere: plain sums)
  s3 = group_sum<LPJ>(s3); sx3 = group_sum<LPJ>(sx3); sy3 = group_sum<LPJ>(sy3);
  sz3 = group_sum<LPJ>(sz3);
  s2 = group_sum<LPJ>(s2); sx2 = group_sum<LPJ>(sx2); sy2 = group_sum<LPJ>(sy2);
  if (joint_ok && li == 0) {
    const double i2 = fast_rcp64(s2), i3 = fast_rcp64(s3);
    const size_t o = (size_t)b * J + j;
    coords2d[o * 2 + 0] = heatmap_to_px(axis_coord_rcp(sx2, i2, ai.w), hs);
    coords2d[o * 2 + 1] = heatmap_to_px(axis_coord_rcp(sy2, i2, ai.h), hs);
    coords3d_rel[o * 3 + 0] = heatmap_to_mm_xy(axis_coord_rcp(sx3, i3, ai.w), hs);
    coords3d_rel[o * 3 + 1] = heatmap_to_mm_xy(axis_coord_rcp(sy3, i3, ai.h), hs);
    coords3d_rel[o * 3 + 2] = heatmap_to_mm_z(axis_coord_rcp(sz3, i3, ai.d), hs);
  }
}

template <typename T, int VEC, int LPJ, int AUX>
static int launch_decode_aux(const void* logits, int B, int J, int D, int H, int W,
                             const HeadScale& hs, float* c2d, float* c3d, hipStream_t stream) {
  constexpr int JPW = kWave / LPJ;
  // consecutive (crop, joint) pairs per wave when joints do not fill a crop's last wave and the tensor is one
  // descriptor's worth (32-bit byte offsets)
  const unsigned long long bytes = (unsigned long long)B * J * (1 + D) * H * W * sizeof(T);
  const bool span = J % JPW != 0 && bytes < 0xffffffffull;
  const long long waves = span ? ((long long)B * J + JPW - 1) / JPW : (long long)B * ((J + JPW - 1) / JPW);
  const int waves_per_block = 4;
  const long long blocks = (waves + waves_per_block - 1) / waves_per_block;
  if (blocks > 0x7fffffffLL) return MTR_E_SHAPE;
  MTR_CLEAR_STALE();
  if (span)
    hipLaunchKernelGGL((decode_nchw_kernel<T, VEC, LPJ, AUX, true>), dim3((unsigned)blocks), dim3(256), 0,
                       stream, (const T*)logits, B, J, D, H, W, hs, make_axis_inv(W, H, D), c2d, c3d);
  else
    hipLaunchKernelGGL((decode_nchw_kernel<T, VEC, LPJ, AUX, false>), dim3((unsigned)blocks), dim3(256), 0,
                       stream, (const T*)logits, B, J, D, H, W, hs, make_axis_inv(W, H, D), c2d, c3d);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

template <typename T, int VEC, int LPJ>
static int launch_decode(const void* logits, int B, int J, int D, int H, int W, const HeadScale& hs,
                         float* c2d, float* c3d, hipStream_t stream) {
  // Non-temporal loads pay when every wave-wide load covers whole 128-B lines (8x8 f32 rows are
  // 256 B: +3 % at D=8, 73 -> 83 % of HBM at D=72); when map rows straddle lines (12x12: 576 B)
  // the neighbouring load re-fetches the evicted line (69 -> 58 %), so those keep the default policy.
  const bool whole_lines = ((size_t)H * W * sizeof(T)) % 128 == 0 && VEC * sizeof(T) == 16;
  if (whole_lines) return launch_decode_aux<T, VEC, LPJ, 2>(logits, B, J, D, H, W, hs, c2d, c3d, stream);
  return launch_decode_aux<T, VEC, LPJ, 0>(logits, B, J, D, H, W, hs, c2d, c3d, stream);
}

template <typename T>
static int dispatch_decode(const void* logits, int B, int J, int D, int H, int W,
                           const HeadScale& hs, float* c2d, float* c3d, hipStream_t stream) {
  const int HW = H * W;
  if constexpr (sizeof(T) == 2) {
    // 16-bit logits (the reference's autocast GPU path): 8 elements = 16 bytes per lane, so that a
    // wave keeps as many BYTES in flight as with f32 (8 x 16-B loads per lane); with 4-element
    // loads the same kernel was latency-starved: 372 us = 3.5 TB/s on the 0.64 GB shape
    const bool vec8 = (W % 8 == 0) && (((uintptr_t)logits) % 16 == 0);
    if (vec8) {
      if (HW <= 64) return launch_decode<T, 8, 8>(logits, B, J, D, H, W, hs, c2d, c3d, stream);
      if (HW <= 512) return launch_decode<T, 8, 16>(logits, B, J, D, H, W, hs, c2d, c3d, stream);
      return launch_decode<T, 8, 64>(logits, B, J, D, H, W, hs, c2d, c3d, stream);
    }
  }
  const bool vec4 = (W % 4 == 0) && (((uintptr_t)logits) % (4 * sizeof(T)) == 0);
  if (vec4) {
    // 16 lanes x 4 elements cover 64 positions per round; wider maps use the whole wave per joint
    if (HW <= 256) return launch_decode<T, 4, 16>(logits, B, J, D, H, W, hs, c2d, c3d, stream);
    return launch_decode<T, 4, 64>(logits, B, J, D, H, W, hs, c2d, c3d, stream);
  }
  if (HW <= 64) return launch_decode<T, 1, 16>(logits, B, J, D, H, W, hs, c2d, c3d, stream);
  return launch_decode<T, 1, 64>(logits, B, J, D, H, W, hs, c2d, c3d, stream);
}

// ---- NHWC logits [B, H, W, J*(1+D)] (the TF twin's layout, 'b h w (d j)',
// metrabs_tf/models/metrabs.py:100-101): a position's channels are contiguous, so lanes walk the
// CHANNELS (one wave-wide load = 256 consecutive bytes of a position) and every lane keeps the online
// softmax state of its own channel row over the positions: running max (f32), sum of e, sum of e*w,
// sum of e*h (f64).  The rows of a joint -- its 2D row n = j and its depth slices n = J + d*J + j,
// J channels apart -- then meet in LDS, where one thread per joint merges them like an online
// softmax over slices.  One workgroup per crop; every logit is read once, coalesced.
// Round 3: G = blockDim / N position GROUPS per channel run side by side (thread t: channel t % N,
// group t / N takes positions g, g + G, ...) and their online-softmax states are merged in LDS in
// group order -- at J = 17, D = 8 (N = 153) a 1024-thread workgroup keeps 918 lanes busy on ~11
// positions each instead of 153 lanes on 64.  N > blockDim: one group, channels in rounds.
// Batches that leave CUs idle (B < 256) deal a crop's JOINTS to `splits` workgroups: a joint's rows
// (its 2D row and its D depth slices) stay in one workgroup, so nothing is merged across workgroups;
// workgroup (b, k) walks the channels {j} and {J + d J + j} of its joints j0 <= j < j1 (1 + D runs of
// j1 - j0 consecutive channels per position) with N = (j1 - j0)(1 + D) local rows.
// The walk of ONE channel over ALL positions in map order when the whole workgroup walks together (G = 1) and
// the map rows are whole batches of U positions (W % U == 0; U = 4, 8, or the row itself: 12, 16): a batch then lies in one map row,
// the positions are wave-uniform (row h and first column w0 of the batch live in scalars) and the moment
// sums factor -- per logit one f64 add (sum e) and one f64 fma with a compile-time column offset (sum u e);
// per batch  sx += su + w0 s,  sy += h s.  Same running-maximum rule (one f64 rescale per batch that raises
// it) and the same f64 accumulators as the generic walk: 11 VALU slots per logit instead of ~20 (the generic
// walk recomputes row and column of every position in every lane and converts both to f64).  Round 5.
#ifndef MTR_NHWC_ONE_KERNEL
#define MTR_NHWC_ONE_KERNEL 0   // 1 (developer A/B builds): one kernel holding all four row walks, as in round 5
#endif
#ifndef MTR_NHWC_PREFETCH
#define MTR_NHWC_PREFETCH 0   // 1: the next batch of loads requested before the current one is summed (round 6: measured SLOWER,
                              // 288 vs 272 us on the 1.28 GB shape, 308 vs 255 us with 16-bit logits -- profiles/r06c_nhwc_decode_ab.jsonl;
                              // the walk is not bound by its dependent round trips)
#endif
// One lane's online-softmax state of its channel over the map, advanced a batch of U positions of ONE map row at a time
// (the factored walk described above).  Both NHWC kernels -- the one that loads its logits from global memory and the
// one that reads them out of a staged copy in LDS -- advance this state with the same calls in the same order: the
// same bits.
template <typename T, int U>
struct NhwcRowWalk {
  float m = -INFINITY;                 // the level q of the sums below (an integer, or -inf: nothing finite yet)
  double s = 0.0, sx = 0.0, sy = 0.0;  // sum of 2^(x log2 e - q), ... times column, ... times row
  int h = 0, w0 = 0;  // (wave-uniform: functions of the batch counter alone)
  __device__ __forceinline__ void batch(const T (&raw)[U], int W) {
    float v[U];
    float mb = -INFINITY;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = to_f32(raw[u]);
      mb = fmaxf(mb, v[u]);
    }
    // The reference level of the weights is a POWER OF TWO: m = q = ceil(max * log2 e), an integer, and a logit weighs
    // exp2(x log2 e - q) (the product exact inside the fma, one rounding; the largest weights lie in (1/2, 1]).  Raising q
    // rescales what has been summed by 2^(q_old - q_new) -- an exponent shift (v_ldexp_f64), EXACT, a handful of
    // instructions -- where the walk used to evaluate exp(m_old - m_new) in f64 (a degree-11 polynomial: ~45 VALU
    // instructions that a wave paid in nearly every batch, since one lane out of 64 raising its maximum is enough).
    const float t = mb * kLog2e;
    if (t > m) {
      const float qn = ceilf(t);
      if (m != -INFINITY) {
        const int d = (int)(m - qn);  // (integers: exact; saturates far below the underflow of the sums)
        s = ldexp(s, d); sx = ldexp(sx, d); sy = ldexp(sy, d);
      }
      m = qn;
    }
    // A -inf logit weighs nothing: under a finite level fma(-inf, log2e, -q) = -inf and exp2(-inf) = +0 by itself; -inf
    // - -inf would be NaN under a -inf level (nothing finite yet, this batch included): the shift is 0 there and every
    // weight of the batch +0 again.  (One select per batch; it was a compare and two selects per logit.)
    const float nm = m == -INFINITY ? 0.0f : -m;
    double sb = 0.0, su = 0.0;
    if constexpr (sizeof(T) == 2) {
      // 16-bit logits (11 significant bits in): the batch's two sums in f32 -- at most 16 terms in (0, 1] -- promoted to
      // f64 once per batch, as the NCHW kernel sums a position's depth slices: two f32 operations per logit instead of
      // a conversion and two f64 ones (the walk is VALU-bound on 16-bit logits: same time per logit as f32 ones)
      float sbf = 0.0f, suf = 0.0f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float e = exp_shifted(v[u], nm);
        sbf += e;
        suf = fmaf(e, (float)u, suf);
      }
      sb = (double)sbf;
      su = (double)suf;
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double e = (double)exp_shifted(v[u], nm);
        sb += e;
        su = fma(e, (double)u, su);
      }
    }
    s += sb;
    sx += fma((double)w0, sb, su);
    sy = fma((double)h, sb, sy);
    w0 += U;
    if (w0 >= W) { w0 = 0; ++h; }
  }
};

template <typename T, int U>
__device__ __forceinline__ void nhwc_walk_rows(const T* __restrict__ x, int ch, int NC, int HW, int W, float& m_out,
                                               double& s_out, double& sx_out, double& sy_out) {
  NhwcRowWalk<T, U> st;
  const T* xp = x + ch;
  // MTR_NHWC_PREFETCH (round 6, a developer option): the NEXT batch's U loads are issued before this batch is summed
  // (two batches = 2 U loads per lane in flight).  Same values, same order of every sum: the same bits -- and slower
  // (see the macro): the extra registers cost more resident workgroups than the deeper queue buys.
  T raw[U], nxt[U];
#pragma unroll
  for (int u = 0; u < U; ++u) raw[u] = xp[(size_t)u * NC];
  for (int p0 = 0; p0 < HW; p0 += U) {
    if (MTR_NHWC_PREFETCH && p0 + U < HW) {  // (uniform)
#pragma unroll
      for (int u = 0; u < U; ++u) nxt[u] = xp[(size_t)(p0 + U + u) * NC];
    }
    st.batch(raw, W);
    if (MTR_NHWC_PREFETCH) {
#pragma unroll
      for (int u = 0; u < U; ++u) raw[u] = nxt[u];
    } else if (p0 + U < HW) {
#pragma unroll
      for (int u = 0; u < U; ++u) raw[u] = xp[(size_t)(p0 + U + u) * NC];
    }
  }
  m_out = st.m; s_out = st.s; sx_out = st.sx; sy_out = st.sy;
}

// (Round 6, measured and removed: FOUR CHANNELS PER LANE -- thread (g, q) loads the 16 bytes of channels 4 q .. 4 q + 3 at
// the eight positions g, g + G, ... in one burst (a wave-wide load ~1 KiB of consecutive memory, as in the NCHW kernel),
// maxima in registers, the G = H*W / 8 partial states of a channel merged in LDS: 495 vs 270 us on the 1.28 GB shape,
// 409 vs 138 us on 12x12 maps (profiles/r06n_nhwc_quad_ab.jsonl) -- 34 - 77 KB of partial states per workgroup leave four
// workgroups per CU, each of which loads for a third of its life and merges 8 - 18 x the states.)
// The tail of the NHWC kernel: [G][N] partial online-softmax states in LDS (row_m: running maxima, row_s: (s, sx, sy)
// in f64) -> merged per channel in group order, scaled per joint over its slices, summed in slice order, stored.
__device__ __forceinline__ void nhwc_merge_and_store(float* row_m, double* row_s, int G, int N, int nj, int j0, int J,
                                                     int D, int b, const HeadScale& hs, const AxisInv& ai,
                                                     float* __restrict__ coords2d, float* __restrict__ coords3d_rel) {
  __syncthreads();
  // The merges -- the groups of a channel, then the slices of a joint -- are online-softmax merges: every partial
  // sum is brought to the common level and added in order.  Levels are integers (powers of two, NhwcRowWalk::batch):
  // the factor 2^(its level - the common level) is an exponent shift, exact (round 6; it was exp(difference of
  // maxima) evaluated in f64, a degree-11 polynomial per partial).  One thread per PARTIAL shifts (all G N of
  // them side by side, then all N), one thread per channel / per joint adds in order.
  const float* chan_m = row_m;  // level of channel n over all its positions
  if (G > 1) {
    float* cm = reinterpret_cast<float*>(row_s + (size_t)G * N * 3);  // [N]
    for (int t = threadIdx.x; t < G * N; t += blockDim.x) {
      const int n = t % N;
      float M = row_m[n];
      for (int g = 1; g < G; ++g) M = fmaxf(M, row_m[g * N + n]);
      const int d = row_m[t] == -INFINITY ? 0 : (int)(row_m[t] - M);  // (a -inf level: sums of 0)
      row_s[t * 3 + 0] = ldexp(row_s[t * 3 + 0], d); row_s[t * 3 + 1] = ldexp(row_s[t * 3 + 1], d);
      row_s[t * 3 + 2] = ldexp(row_s[t * 3 + 2], d);
      if (t < N) cm[n] = M;
    }
    chan_m = cm;
    __syncthreads();
  }
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    double S = row_s[n * 3], SX = row_s[n * 3 + 1], SY = row_s[n * 3 + 2];
    for (int g = 1; g < G; ++g) {  // group order
      const int t = g * N + n;
      S += row_s[t * 3]; SX += row_s[t * 3 + 1]; SY += row_s[t * 3 + 2];
    }
    const int slice = n / nj, jj = n - slice * nj;
    if (slice > 0) {  // a depth slice: scaled to the maximum over its joint's slices
      float M = -INFINITY;
      for (int d = 0; d < D; ++d) M = fmaxf(M, chan_m[nj + d * nj + jj]);
      const int d = chan_m[n] == -INFINITY ? 0 : (int)(chan_m[n] - M);
      S = ldexp(S, d); SX = ldexp(SX, d); SY = ldexp(SY, d);
    }
    row_s[n * 3 + 0] = S; row_s[n * 3 + 1] = SX; row_s[n * 3 + 2] = SY;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < nj; j += blockDim.x) {
    const size_t o = (size_t)b * J + j0 + j;
    {
      const double i2 = fast_rcp64(row_s[j * 3]);
      coords2d[o * 2 + 0] = heatmap_to_px(axis_coord_rcp(row_s[j * 3 + 1], i2, ai.w), hs);
      coords2d[o * 2 + 1] = heatmap_to_px(axis_coord_rcp(row_s[j * 3 + 2], i2, ai.h), hs);
    }
    double S = 0.0, SX = 0.0, SY = 0.0, SZ = 0.0;
    for (int d = 0; d < D; ++d) {  // slice order
      const int n = nj + d * nj + j;
      const double sd = row_s[n * 3];
      S += sd; SX += row_s[n * 3 + 1]; SY += row_s[n * 3 + 2]; SZ = fma(sd, (double)d, SZ);
    }
    const double i3 = fast_rcp64(S);
    coords3d_rel[o * 3 + 0] = heatmap_to_mm_xy(axis_coord_rcp(SX, i3, ai.w), hs);
    coords3d_rel[o * 3 + 1] = heatmap_to_mm_xy(axis_coord_rcp(SY, i3, ai.h), hs);
    coords3d_rel[o * 3 + 2] = heatmap_to_mm_z(axis_coord_rcp(SZ, i3, ai.d), hs);
  }
}

// RB: the map-row batch this instantiation's factored walk is compiled for (4, 8, 12, 16; 0 = none): one walk per
// instantiation instead of all four in one kernel (round 6) -- the register allocation of a launch is its own
// walk's, not the 16-position one's.
template <typename T, int RB>
__global__ __launch_bounds__(1024) void decode_nhwc_kernel(const T* __restrict__ logits, int B, int J,
                                                           int D, int H, int W, int splits, HeadScale hs,
                                                           AxisInv ai, float* __restrict__ coords2d,
                                                           float* __restrict__ coords3d_rel) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int NC = J * (1 + D), HW = H * W;  // channels per position
  const int b = blockIdx.x / splits, part = blockIdx.x - b * splits;
  const int j0 = (int)((long long)part * J / splits), j1 = (int)((long long)(part + 1) * J / splits);
  const int nj = j1 - j0, N = nj * (1 + D);  // this workgroup's joints / rows
  // position groups per channel: as many as the threads allow, but a group should have ~8 positions
  // to walk (one batch of loads) and the groups of a channel are merged one after the other
  int G = N <= (int)blockDim.x ? (int)blockDim.x / N : 1;
  G = max(1, min(G, min(16, HW / 8)));
  float* row_m = reinterpret_cast<float*>(smem_raw);                               // [G][N]
  double* row_s = reinterpret_cast<double*>(smem_raw + ((G * N * 4 + 15) & ~15));  // [G][N][3]
  const T* x = logits + (size_t)b * HW * NC;
  constexpr int U = 8;  // positions per batch: their loads are in flight together (a walk of one
                        // dependent load per step was the whole 12 us of a 64-crop launch)
  const float rcp_w = __frcp_rn((float)W);
  // (workgroup-uniform) a whole map row per batch for the usual widths, else 8 or 4 positions of a row
  // (the host instantiates RB by the same rule; RB = -1, developer A/B builds: round 5's one kernel with all four walks)
  const int row_batch = G != 1 ? 0 : RB >= 0 ? RB : (W == 12 || W == 16) ? W : (W % 8 == 0 ? 8 : (W % 4 == 0 ? 4 : 0));
  for (int t = threadIdx.x; t < G * N; t += blockDim.x) {
    const int n = t % N, g = t / N;
    // local row n = (slice, joint): slice 0 is the 2D row, slice 1 + d the depth slice d
    const int slice = n / nj, jj = n - slice * nj;
    const int ch = slice == 0 ? j0 + jj : J + (slice - 1) * J + j0 + jj;
    float m = -INFINITY;
    double s = 0.0, sx = 0.0, sy = 0.0;
    if constexpr (RB != 0) {
      if (row_batch) {
        if constexpr (RB > 0) {
          nhwc_walk_rows<T, RB>(x, ch, NC, HW, W, m, s, sx, sy);
        } else {
          if (row_batch == 8) nhwc_walk_rows<T, 8>(x, ch, NC, HW, W, m, s, sx, sy);
          else if (row_batch == 12) nhwc_walk_rows<T, 12>(x, ch, NC, HW, W, m, s, sx, sy);
          else if (row_batch == 16) nhwc_walk_rows<T, 16>(x, ch, NC, HW, W, m, s, sx, sy);
          else nhwc_walk_rows<T, 4>(x, ch, NC, HW, W, m, s, sx, sy);
        }
        row_m[t] = m;
        row_s[t * 3 + 0] = s; row_s[t * 3 + 1] = sx; row_s[t * 3 + 2] = sy;
        continue;
      }
    }
    for (int p0 = g; p0 < HW; p0 += G * U) {
      float v[U];
      float mb = -INFINITY;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = p0 + u * G;
        v[u] = p < HW ? to_f32(x[(size_t)p * NC + ch]) : -INFINITY;
        mb = fmaxf(mb, v[u]);
      }
      const float tq = mb * kLog2e;
      if (tq > m) {  // raise the power-of-two level of the sums (NhwcRowWalk::batch): an exact exponent shift
        const float qn = ceilf(tq);
        if (m != -INFINITY) {  // (nothing summed yet otherwise)
          const int d = (int)(m - qn);
          s = ldexp(s, d); sx = ldexp(sx, d); sy = ldexp(sy, d);
        }
        m = qn;
      }
      const float nm = -m;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = p0 + u * G;
        // a -inf logit (and a position past the map) weighs nothing (-inf - -inf is NaN under a -inf maximum)
        if (v[u] != -INFINITY) {
          // (row / column of the position without an integer division per logit: exact for p < 2^16, see
          //  head_rt.hip rt_decode_groups; the division was most of this VALU-bound loop)
          const int h = HW <= 65536 ? (int)(((float)p + 0.5f) * rcp_w) : p / W, w = p - h * W;
          const double e = (double)exp_shifted(v[u], nm);
          s += e; sx += e * (double)w; sy += e * (double)h;
        }
      }
    }
    row_m[t] = m;
    row_s[t * 3 + 0] = s; row_s[t * 3 + 1] = sx; row_s[t * 3 + 2] = sy;
  }
  nhwc_merge_and_store(row_m, row_s, G, N, nj, j0, J, D, b, hs, ai, coords2d, coords3d_rel);
}

// ---- NHWC, staged (round 6).  The walk above requests its logits as U dword loads per lane (a wave: U x 256 bytes that
// straddle 128-byte lines -- a position's 153 channels are 612 bytes) and has nothing in flight while it sums.  Here the
// logits of a crop reach LDS through a RING of batch slots filled by global_load_lds_dwordx4 -- 16 bytes per lane,
// consecutive lanes consecutive addresses (a crop is one contiguous run), no registers, R - 1 batches in flight per
// workgroup WHILE it sums -- and the same walk (NhwcRowWalk: the same calls in the same order, the same bits) reads its
// channel out of LDS: lane n reads dword p*N + n of a slot, consecutive lanes consecutive banks.  One workgroup per
// crop, one lane per channel, one barrier per batch: [own copies of batch k landed] barrier [copies of batch k + R - 1
// issued into the slot batch k - 1 left] [batch k summed].  A batch's copy starts at the 16-byte granule that holds its
// first byte.  What it buys (the walk itself is instruction-bound -- what moved the large f32
// launches was the walk's own arithmetic): launches of a few rounds (1,024 crops 16.3 -> 9.4 us), 16-bit logits
// (183 -> 165 us on 0.64 GB), 2 - 3 % on the 1.28 GB f32 shape.
// Slots R of the ring, measured on one MI355X (profiles/r06r_nhwc_staged_ab*.jsonl, R = 2 / 3 / 4; 6 is slower
// everywhere): f32 logits in launches that run for many rounds (32,768 crops of 8x8x153: 230 / 222 / 219 us) want
// three batches in flight, everything else (16-bit logits; launches of one or two rounds: 1,024 crops 9.4 / 10.3 /
// 10.4 us; one-wave workgroups) the shortest prologue and the most workgroups per CU -- nhwc_ring_slots.  The bits do
// not depend on R.
__device__ __forceinline__ void nhwc_dma16(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :
               : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(sbase)
               : "memory", "m0");  // (M0 is overwritten: the register allocator must know)
}
template <typename T>
__device__ __forceinline__ T nhwc_lds_at(const __attribute__((address_space(3))) char* p) {
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, *reinterpret_cast<const __attribute__((address_space(3))) unsigned*>(p));
  } else {
    return __builtin_bit_cast(T, *reinterpret_cast<const __attribute__((address_space(3))) unsigned short*>(p));
  }
}
// at most n of this wave's vector-memory operations still outstanding (n wave-uniform; the instruction takes an immediate)
__device__ __forceinline__ void nhwc_wait_vmcnt_le(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: case 10: case 11: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;  // (stricter than asked: safe)
  }
}

// The staged kernel's tail: the states are still in their lanes' registers (lane t = channel n of the workgroup's crop
// cl).  The arithmetic of nhwc_merge_and_store with one position group -- levels through LDS, a depth slice shifted to its
// joint's level, the slices of a joint summed in slice order from 0.0, the same reciprocal and scalings -- with the four
// sums of a joint (S, SX, SY, SZ) in FOUR neighbouring lanes instead of one lane's four chains: the same operations per
// accumulator in the same order, the same bits.
template <int CPW>
__device__ __forceinline__ void nhwc_staged_tail(char* smem_raw, int t, bool active, int cl, int n, float m, double s, double sx,
                                                 double sy, int N, int J, int D, int b0, int B, const HeadScale& hs,
                                                 const AxisInv& ai, float* __restrict__ coords2d,
                                                 float* __restrict__ coords3d_rel) {
  float* row_m = reinterpret_cast<float*>(smem_raw);                                 // [CPW][N]
  double* row_s = reinterpret_cast<double*>(smem_raw + ((CPW * N * 4 + 15) & ~15));  // [CPW][N][3]
  __syncthreads();  // every walk has left the ring: the states below overlay it
  if (active) row_m[t] = m;
  __syncthreads();
  if (active) {
    // (n / J without an integer division: exact for n < 2^16 -- N <= 1,024 here)
    const int slice = (int)(((float)n + 0.5f) * __frcp_rn((float)J)), jj = n - slice * J;
    if (slice > 0) {  // a depth slice: to the level of its joint's slices
      const float* lv = row_m + cl * N + J + jj;
      float M = -INFINITY;
      for (int d = 0; d < D; ++d) M = fmaxf(M, lv[d * J]);
      const int sh = m == -INFINITY ? 0 : (int)(m - M);
      s = ldexp(s, sh); sx = ldexp(sx, sh); sy = ldexp(sy, sh);
    }
    row_s[t * 3 + 0] = s; row_s[t * 3 + 1] = sx; row_s[t * 3 + 2] = sy;
  }
  __syncthreads();
  // lane 4 (c J + j) + q: quantity q of joint j of crop c
  if (t < CPW * J * 4) {
    const int q = t & 3, cj = t >> 2;
    const int c = CPW == 1 ? 0 : (cj >= J ? 1 : 0), j = cj - c * J;
    const double* rs = row_s + (size_t)c * N * 3;
    double acc = 0.0;
    const int col = q == 3 ? 0 : q;
    for (int d = 0; d < D; ++d) {  // slice order
      const double v = rs[(J + d * J + j) * 3 + col];
      acc = q == 3 ? fma(v, (double)d, acc) : acc + v;
    }
    // quad lanes 1 .. 3 hand their sums to lane 0 (the 3D coordinates); lane 1 writes the 2D ones meanwhile
    const double SX = dpp_move<kDppQuadBcast1>(acc), SY = dpp_move<kDppQuadBcast2>(acc), SZ = dpp_move<kDppQuadBcast3>(acc);
    const size_t o = (size_t)(b0 + c) * J + j;
    if (b0 + c < B) {
      if (q == 0) {
        const double i3 = fast_rcp64(acc);
        coords3d_rel[o * 3 + 0] = heatmap_to_mm_xy(axis_coord_rcp(SX, i3, ai.w), hs);
        coords3d_rel[o * 3 + 1] = heatmap_to_mm_xy(axis_coord_rcp(SY, i3, ai.h), hs);
        coords3d_rel[o * 3 + 2] = heatmap_to_mm_z(axis_coord_rcp(SZ, i3, ai.d), hs);
      } else if (q == 1) {
        const double i2 = fast_rcp64(rs[j * 3]);
        coords2d[o * 2 + 0] = heatmap_to_px(axis_coord_rcp(rs[j * 3 + 1], i2, ai.w), hs);
        coords2d[o * 2 + 1] = heatmap_to_px(axis_coord_rcp(rs[j * 3 + 2], i2, ai.h), hs);
      }
    }
  }
}

// CPW crops per workgroup (1 or 2): at 153 channels one crop is 64 + 64 + 25 lanes (80 % of three waves), two are 306 of
// 320 (96 % of five) and share one tail.
template <typename T, int U, int R, int CPW>
__global__ __launch_bounds__(1024) void decode_nhwc_staged_kernel(const T* __restrict__ logits, int B, int J, int D,
                                                                  int H, int W, unsigned slot_bytes, int bps, HeadScale hs,
                                                                  AxisInv ai, float* __restrict__ coords2d,
                                                                  float* __restrict__ coords3d_rel) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int N = J * (1 + D), HW = H * W;
  const int b0 = blockIdx.x * CPW;
  // a ring slot holds `bps` batches of U positions (one copy, one wait, one barrier per slot; the walk still advances a
  // batch -- a run of one map row -- at a time: the same calls in the same order whatever bps is)
  const int n_batches = HW / (U * bps);   // slots to walk
  const unsigned pitch = (unsigned)N * (unsigned)sizeof(T);  // bytes between positions
  const unsigned walk_bytes = (unsigned)U * pitch;            // one batch of the walk
  const unsigned batch_bytes = walk_bytes * (unsigned)bps;    // one slot
  const size_t crop_bytes = (size_t)HW * pitch;
  const char* crop0 = uniform_ptr(reinterpret_cast<const char*>(logits) + (size_t)b0 * crop_bytes);
  const unsigned lo0 = (unsigned)(reinterpret_cast<uintptr_t>(crop0) & 15u);  // (wave-uniform)
  const unsigned lo1 = (unsigned)((reinterpret_cast<uintptr_t>(crop0) + crop_bytes) & 15u);
  const auto lds = (const __attribute__((address_space(3))) char*)smem_raw;
  const unsigned lds0 = (unsigned)(size_t)lds;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63,
                 nw = blockDim.x >> 6;
  // the copies of batch k: per crop the granules [start - shift, start + batch_bytes) -> slot k % R, part c; returns this
  // wave's count
  auto issue = [&](int k) -> int {
    int count = 0;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      if (b0 + c >= B) break;  // (uniform: the last workgroup of an odd batch)
      const unsigned shift = ((c ? lo1 : lo0) + (unsigned)k * batch_bytes) & 15u;
      const char* src = crop0 + (size_t)c * crop_bytes + (size_t)k * batch_bytes - shift;
      const unsigned total = shift + batch_bytes;
      const unsigned slot = lds0 + ((unsigned)(k % R) * CPW + (unsigned)c) * slot_bytes;
      // (the second crop's pieces start one wave further: the waves share the odd piece counts)
      for (unsigned q = ((wave + nw - (unsigned)c) % nw) * 1024u; q < total; q += nw * 1024u) {  // (wave-uniform trip count)
        const unsigned off = q + lane * 16u;
        // (the last granule may reach <= 15 bytes past the batch: the same 16-byte granule, the same page)
        if (off < total) nhwc_dma16(src, off, slot + q);
        ++count;
      }
    }
    return count;
  };
  int ahead[R - 1];  // this wave's copy counts of the batches in flight, oldest first
#pragma unroll
  for (int r = 0; r < R - 1; ++r) ahead[r] = r < n_batches ? issue(r) : 0;

  const int t = threadIdx.x;
  const int cl = CPW == 1 ? 0 : (t >= N ? 1 : 0), n = t - cl * N;
  const bool active = t < CPW * N && b0 + cl < B;
  const unsigned mine = (unsigned)cl * slot_bytes + (unsigned)n * (unsigned)sizeof(T);
  const unsigned my_lo = cl ? lo1 : lo0;
  NhwcRowWalk<T, U> st;
  for (int k = 0; k < n_batches; ++k) {
    int later = 0;
#pragma unroll
    for (int r = 1; r < R - 1; ++r) later += ahead[r];
    nhwc_wait_vmcnt_le(__builtin_amdgcn_readfirstlane(later));  // this wave's copies of batch k have landed ...
    __syncthreads();            // ... everyone's have, and everyone has left batch k - 1
#pragma unroll
    for (int r = 0; r + 1 < R - 1; ++r) ahead[r] = ahead[r + 1];
    ahead[R - 2] = k + R - 1 < n_batches ? issue(k + R - 1) : 0;
    if (active) {
      const unsigned shift = (my_lo + (unsigned)k * batch_bytes) & 15u;
      const auto from = lds + (unsigned)(k % R) * CPW * slot_bytes + shift + mine;
      for (int bb = 0; bb < bps; ++bb) {
        T raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) raw[u] = nhwc_lds_at<T>(from + (unsigned)bb * walk_bytes + (unsigned)u * pitch);
        st.batch(raw, W);
      }
    }
  }
  nhwc_staged_tail<CPW>(smem_raw, t, active, cl, n, st.m, st.s, st.sx, st.sy, N, J, D, b0, B, hs, ai, coords2d, coords3d_rel);
}

// The staged kernel's shapes: one lane per channel, the factored walk (W a multiple of 4), a ring of at most 48 KiB.
#ifndef MTR_NHWC_BPS
#define MTR_NHWC_BPS 2   // batches per ring slot for 16-BIT logits, where the map has an even number of them and the ring
                         // still fits: half the barriers and copy bookkeeping, same bits -- 0.64 GB of f16 logits 156.0 ->
                         // 151.1 us, 1,024 crops 8.6 -> 8.1; f32 logits want the smaller ring (32,768 crops 233 -> 251 us with
                         // two, 8,192 x 36 channels 50.5 -> 58.7; profiles/r06zd_nhwc_bps.jsonl) and keep one
#endif
inline size_t nhwc_slot_bytes(long long N, int U, size_t elem) { return ((size_t)U * N * elem + 30 + 15) & ~(size_t)15; }
inline int nhwc_batches_per_slot(long long N, int HW, int U, int R, int cpw, size_t elem) {
  return MTR_NHWC_BPS == 2 && elem == 2 && (HW / U) % 2 == 0 && nhwc_slot_bytes(N, 2 * U, elem) * R * cpw <= 48 * 1024 ? 2 : 1;
}
inline int nhwc_ring_slots(long long B, long long N, size_t elem) { return elem == 4 && B >= 8192 && N >= 128 ? 4 : 2; }
inline bool nhwc_staged_fits(long long N, int W, int U, int R, size_t elem) {
  return N <= 1024 && W % 4 == 0 && nhwc_slot_bytes(N, U, elem) * R <= 48 * 1024;
}

// Two crops per workgroup (nhwc_staging = 3; measured, never the library's choice): where that fills the waves better
// (153 channels: 306 lanes of 320 instead of 153 of 192), fits the ring and still leaves a workgroup per CU.  Same bits;
// 32,768 crops of 8x8x153 218.9 -> 216.8 us f32, 166.0 -> 164.9 f16, but 4,096 crops 28.2 -> 29.5 and 8,192 crops of 216
// f16 channels 52.9 -> 57.7 (profiles/r06y_nhwc_cpw.jsonl): five waves at one barrier per batch lose what the fuller
// last wave and the shared tail buy.
inline int nhwc_crops_per_wg(long long B, long long N, int U, int R, size_t elem) {
  const long long waste1 = (N + 63) / 64 * 64 - N, waste2 = (2 * N + 63) / 64 * 64 - 2 * N;
  return 2 * N <= 1024 && waste2 < 2 * waste1 && B >= 512 && nhwc_slot_bytes(N, U, elem) * R * 2 <= 48 * 1024 ? 2 : 1;
}
template <typename T, int R, int CPW>
static void launch_decode_nhwc_staged(const void* logits, int B, int J, int D, int H, int W, int rb, const HeadScale& hs,
                                      float* c2d, float* c3d, hipStream_t stream) {
  const long long N = (long long)J * (1 + D);
  auto kern = rb == 16 ? decode_nhwc_staged_kernel<T, 16, R, CPW> : rb == 12 ? decode_nhwc_staged_kernel<T, 12, R, CPW>
              : rb == 8 ? decode_nhwc_staged_kernel<T, 8, R, CPW> : decode_nhwc_staged_kernel<T, 4, R, CPW>;
  const int bps = nhwc_batches_per_slot(N, H * W, rb, R, CPW, sizeof(T));
  const size_t slot = nhwc_slot_bytes(N, rb * bps, sizeof(T));
  const size_t ring = slot * R * CPW;
  const size_t tail = (size_t)((CPW * N * 4 + 15) & ~15LL) + (size_t)CPW * N * 24;
  const size_t lds = ring > tail ? ring : tail;
  const int threads = (int)((CPW * N + 63) / 64 * 64);
  hipLaunchKernelGGL(kern, dim3((unsigned)((B + CPW - 1) / CPW)), dim3(threads), lds, stream, (const T*)logits, B, J, D, H,
                     W, (unsigned)slot, bps, hs, make_axis_inv(W, H, D), c2d, c3d);
}

template <typename T>
static int launch_decode_nhwc(const void* logits, int B, int J, int D, int H, int W, const HeadScale& hs,
                              float* c2d, float* c3d, hipStream_t stream, int staging = 0) {
  const long long N = (long long)J * (1 + D);
  // staging: 0 = the library's rule (launches of >= 256 crops), 1 = never, 2 = whenever the shape allows, 3 = as 2 with
  // two crops per workgroup where that fills the waves better
  {
    const int rb = (W == 12 || W == 16) ? W : (W % 8 == 0 ? 8 : 4);
    const int R = nhwc_ring_slots(B, N, sizeof(T));
    if (staging != 1 && nhwc_staged_fits(N, W, rb, R, sizeof(T)) && (staging >= 2 || B >= 256)) {
      MTR_CLEAR_STALE();
      const int cpw = staging == 3 ? nhwc_crops_per_wg(B, N, rb, R, sizeof(T)) : 1;
      if (R == 4 && cpw == 2) launch_decode_nhwc_staged<T, 4, 2>(logits, B, J, D, H, W, rb, hs, c2d, c3d, stream);
      else if (R == 4) launch_decode_nhwc_staged<T, 4, 1>(logits, B, J, D, H, W, rb, hs, c2d, c3d, stream);
      else if (cpw == 2) launch_decode_nhwc_staged<T, 2, 2>(logits, B, J, D, H, W, rb, hs, c2d, c3d, stream);
      else launch_decode_nhwc_staged<T, 2, 1>(logits, B, J, D, H, W, rb, hs, c2d, c3d, stream);
      MTR_CHECK_LAUNCH();
      return MTR_OK;
    }
  }
  // 1024 threads (up to 6 position groups per channel at N = 153) while a crop is ONE workgroup and the
  // batch does not fill the chip; large batches keep 256 threads (4+ workgroups per CU hide the walk)
  // joints of a crop over `splits` workgroups while the batch leaves CUs idle (B = 64: 4 x 64 workgroups)
  int splits = B < 256 ? 256 / B : 1;
  if (splits > J) splits = J;
  // threads: the rows of the largest part x its position groups (<= 16, ~8 positions each), 256..1024
  int threads = 256;
  {
    const long long rows = (long long)((J + splits - 1) / splits) * (1 + D);
    long long groups = (long long)H * W / 8;
    groups = groups < 1 ? 1 : groups > 16 ? 16 : groups;
    // ... and no more threads than keep every workgroup of the launch resident (2048 per CU)
    const long long resident = 2048LL * 256 / ((long long)B * splits) / rows;
    if (groups > resident) groups = resident < 1 ? 1 : resident;
    const long long want = (rows * groups + 63) / 64 * 64;
    // (round 5: whole waves, no 256-thread floor -- N = 153 rows in one group are three waves; the fourth of
    //  a 256-thread workgroup only held a SIMD slot)
    threads = (int)(want < 64 ? 64 : want > 1024 ? 1024 : want);
    if (want > 1024) {  // channels in rounds: even rounds (N = 1098: 2 x 576 lanes, not 1024 + 74)
      const long long rounds = (want + 1023) / 1024;
      threads = (int)(((want + rounds - 1) / rounds + 63) / 64 * 64);
    }
  }
  // (LDS is sized for the unsplit row count: an upper bound of every part's G * N)
  const long long slots = N <= threads ? (long long)threads : N;
  const size_t lds = (size_t)((slots * 4 + 15) & ~15LL) + (size_t)slots * 24 + (size_t)N * 4;
  if (lds > 160 * 1024 - 256) return MTR_E_SHAPE;  // > 5,800 channels per position
  // a whole map row per batch for the usual widths, else 8 or 4 positions of a row (the kernel's rule)
  const int rb = (W == 12 || W == 16) ? W : (W % 8 == 0 ? 8 : (W % 4 == 0 ? 4 : 0));
#if MTR_NHWC_ONE_KERNEL
  auto kern = decode_nhwc_kernel<T, -1>;
  (void)rb;
#else
  auto kern = rb == 16 ? decode_nhwc_kernel<T, 16> : rb == 12 ? decode_nhwc_kernel<T, 12>
              : rb == 8 ? decode_nhwc_kernel<T, 8> : rb == 4 ? decode_nhwc_kernel<T, 4> : decode_nhwc_kernel<T, 0>;
#endif
  if (lds > 64 * 1024) {
    const int rc = allow_dynamic_lds((const void*)kern, lds);
    if (rc != MTR_OK) return rc;
  }
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(kern, dim3((unsigned)(B * splits)), dim3(threads), lds, stream, (const T*)logits, B, J, D,
                     H, W, splits, hs, make_axis_inv(W, H, D), c2d, c3d);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

}  // namespace mtr

extern "C" int mtr_softargmax_decode(const void* logits, int dtype, int layout, int B, int J, int D,
                                     int H, int W, const mtr_head_params* p, float* coords2d,
                                     float* coords3d_rel, mtr_stream_t stream) {
  return mtr_softargmax_decode_opts(logits, dtype, layout, B, J, D, H, W, p, 0, coords2d, coords3d_rel, stream);
}

extern "C" int mtr_softargmax_decode_opts(const void* logits, int dtype, int layout, int B, int J, int D,
                                          int H, int W, const mtr_head_params* p, int nhwc_staging, float* coords2d,
                                          float* coords3d_rel, mtr_stream_t stream) {
  if (nhwc_staging < 0 || nhwc_staging > 3) return MTR_E_PARAM;
  if (!logits || !p || !coords2d || !coords3d_rel) return MTR_E_NULL;
  if (B < 0 || J <= 0 || D <= 0 || H <= 0 || W <= 0) return MTR_E_SHAPE;
  if (p->proc_side <= 0 || p->stride_test <= 0) return MTR_E_PARAM;
  if (layout != MTR_NCHW && layout != MTR_NHWC) return MTR_E_DTYPE;
  if (B == 0) return MTR_OK;
  const mtr::HeadScale hs = mtr::make_head_scale(*p);
  hipStream_t s = (hipStream_t)stream;
  if (layout == MTR_NHWC) {
    switch (dtype) {
      case MTR_F32: return mtr::launch_decode_nhwc<float>(logits, B, J, D, H, W, hs, coords2d, coords3d_rel, s, nhwc_staging);
      case MTR_F16: return mtr::launch_decode_nhwc<__half>(logits, B, J, D, H, W, hs, coords2d, coords3d_rel, s, nhwc_staging);
      case MTR_BF16: return mtr::launch_decode_nhwc<__hip_bfloat16>(logits, B, J, D, H, W, hs, coords2d, coords3d_rel, s, nhwc_staging);
      default: return MTR_E_DTYPE;
    }
  }
  switch (dtype) {
    case MTR_F32: return mtr::dispatch_decode<float>(logits, B, J, D, H, W, hs, coords2d, coords3d_rel, s);
    case MTR_F16: return mtr::dispatch_decode<__half>(logits, B, J, D, H, W, hs, coords2d, coords3d_rel, s);
    case MTR_BF16: return mtr::dispatch_decode<__hip_bfloat16>(logits, B, J, D, H, W, hs, coords2d, coords3d_rel, s);
    default: return MTR_E_DTYPE;
  }
}
