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
//   * per lane an ONLINE softmax (running max m, rescale on growth) so any D (8 .. 72 ..) and any
//     H*W stream through a fixed register budget; D slices are fetched 8 at a time so 8-9
//     independent 16-B loads per lane are in flight;
//   * exp in fp32 (accurate expf), the four moment sums (total, x, y, z) in fp64, merged across the
//     group with xor-butterflies after rescaling each lane to the group max.
#include "common.h"

namespace mtr {

struct Moments3 {
  float m;
  double s, sx, sy, sz;
};
struct Moments2 {
  float m;
  double s, sx, sy;
};

template <int VEC>
__device__ __forceinline__ float vec_max(const float (&v)[VEC]) {
  float r = v[0];
#pragma unroll
  for (int i = 1; i < VEC; ++i) r = fmaxf(r, v[i]);
  return r;
}

template <typename T, int VEC, int LPJ>
__global__ __launch_bounds__(256) void decode_nchw_kernel(
    const T* __restrict__ logits, int B, int J, int D, int H, int W, HeadScale hs,
    float* __restrict__ coords2d, float* __restrict__ coords3d_rel) {
  constexpr int JPW = kWave / LPJ;  // joints per wave
  constexpr int CH = 8;             // depth slices fetched per round
  const int HW = H * W;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
  const int groups_per_crop = (J + JPW - 1) / JPW;
  const int b = wave / groups_per_crop;
  if (b >= B) return;  // wave-uniform
  const int j = (wave % groups_per_crop) * JPW + lane / LPJ;
  const int li = lane % LPJ;
  const bool joint_ok = j < J;

  const size_t chan = (size_t)HW;
  const T* crop = logits + (size_t)b * (size_t)(J * (1 + D)) * chan;
  const T* base2d = crop + (size_t)j * chan;
  const T* base3d = crop + (size_t)(J + j) * chan;  // slice d is d*J channels further
  const size_t dstride = (size_t)J * chan;

  Moments3 a3{-INFINITY, 0.0, 0.0, 0.0, 0.0};
  Moments2 a2{-INFINITY, 0.0, 0.0, 0.0};

  const int n_iter = (HW + VEC * LPJ - 1) / (VEC * LPJ);
  for (int it = 0; it < n_iter; ++it) {
    const int p0 = (it * LPJ + li) * VEC;
    const bool active = joint_ok && p0 < HW;
    // (h, w) of the VEC elements; VEC <= W always holds for the VEC=4 instantiations (W % 4 == 0)
    const int h0 = p0 / W, w0 = p0 - h0 * W;
    float fx[VEC], fy[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      int w = w0 + v, h = h0;
      if (w >= W) { w -= W; ++h; }
      fx[v] = (float)w;
      fy[v] = (float)h;
    }

    // ---- 2D heatmap of this joint (softmax over H*W only)
    float v2[VEC];
    if (active) load_vec<T, VEC>(base2d + p0, v2);
    // ---- first round of depth slices is issued before any arithmetic
    for (int d0 = 0; d0 < D; d0 += CH) {
      float v3[CH][VEC];
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        if (active && d0 + k < D) {
          load_vec<T, VEC>(base3d + (size_t)(d0 + k) * dstride + p0, v3[k]);
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) v3[k][v] = -INFINITY;
        }
      }
      if (d0 == 0 && active) {
        const float cm = vec_max<VEC>(v2);
        if (cm > a2.m) {
          const double r = (double)expf(a2.m - cm);
          a2.s *= r; a2.sx *= r; a2.sy *= r;
          a2.m = cm;
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const double e = (double)expf(v2[v] - a2.m);
          a2.s += e;
          a2.sx += e * (double)fx[v];
          a2.sy += e * (double)fy[v];
        }
      }
      if (active) {
        float cm = vec_max<VEC>(v3[0]);
#pragma unroll
        for (int k = 1; k < CH; ++k) cm = fmaxf(cm, vec_max<VEC>(v3[k]));
        if (cm > a3.m) {
          const double r = (double)expf(a3.m - cm);
          a3.s *= r; a3.sx *= r; a3.sy *= r; a3.sz *= r;
          a3.m = cm;
        }
        // separable accumulation: per (h,w) column the sum over depth, then weight by x / y once
        double col[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) col[v] = 0.0;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const double fz = (double)(d0 + k);
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const double e = (double)expf(v3[k][v] - a3.m);  // exp(-inf) = 0 for padded slices
            col[v] += e;
            a3.sz += e * fz;
          }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          a3.s += col[v];
          a3.sx += col[v] * (double)fx[v];
          a3.sy += col[v] * (double)fy[v];
        }
      }
    }
  }

  // ---- merge the LPJ lanes of the group: rescale to the group max, then sum
  {
    const float gm = group_max<LPJ>(a3.m);
    const double r = (a3.m == -INFINITY) ? 0.0 : (double)expf(a3.m - gm);
    a3.s = group_sum<LPJ>(a3.s * r);
    a3.sx = group_sum<LPJ>(a3.sx * r);
    a3.sy = group_sum<LPJ>(a3.sy * r);
    a3.sz = group_sum<LPJ>(a3.sz * r);
    const float gm2 = group_max<LPJ>(a2.m);
    const double r2 = (a2.m == -INFINITY) ? 0.0 : (double)expf(a2.m - gm2);
    a2.s = group_sum<LPJ>(a2.s * r2);
    a2.sx = group_sum<LPJ>(a2.sx * r2);
    a2.sy = group_sum<LPJ>(a2.sy * r2);
  }
  if (joint_ok && li == 0) {
    const size_t o = (size_t)b * J + j;
    coords2d[o * 2 + 0] = heatmap_to_px(axis_coord(a2.sx, a2.s, W), hs);
    coords2d[o * 2 + 1] = heatmap_to_px(axis_coord(a2.sy, a2.s, H), hs);
    coords3d_rel[o * 3 + 0] = heatmap_to_mm_xy(axis_coord(a3.sx, a3.s, W), hs);
    coords3d_rel[o * 3 + 1] = heatmap_to_mm_xy(axis_coord(a3.sy, a3.s, H), hs);
    coords3d_rel[o * 3 + 2] = heatmap_to_mm_z(axis_coord(a3.sz, a3.s, D), hs);
  }
}

template <typename T, int VEC, int LPJ>
static int launch_decode(const void* logits, int B, int J, int D, int H, int W, const HeadScale& hs,
                         float* c2d, float* c3d, hipStream_t stream) {
  constexpr int JPW = kWave / LPJ;
  const long long waves = (long long)B * ((J + JPW - 1) / JPW);
  const int waves_per_block = 4;
  const long long blocks = (waves + waves_per_block - 1) / waves_per_block;
  if (blocks > 0x7fffffffLL) return MTR_E_SHAPE;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL((decode_nchw_kernel<T, VEC, LPJ>), dim3((unsigned)blocks), dim3(256), 0, stream,
                     (const T*)logits, B, J, D, H, W, hs, c2d, c3d);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

template <typename T>
static int dispatch_decode(const void* logits, int B, int J, int D, int H, int W,
                           const HeadScale& hs, float* c2d, float* c3d, hipStream_t stream) {
  const int HW = H * W;
  const bool vec4 = (W % 4 == 0) && (((uintptr_t)logits) % (4 * sizeof(T)) == 0);
  if (vec4) {
    // 16 lanes x 4 elements cover 64 positions per round; wider maps use the whole wave per joint
    if (HW <= 256) return launch_decode<T, 4, 16>(logits, B, J, D, H, W, hs, c2d, c3d, stream);
    return launch_decode<T, 4, 64>(logits, B, J, D, H, W, hs, c2d, c3d, stream);
  }
  if (HW <= 64) return launch_decode<T, 1, 16>(logits, B, J, D, H, W, hs, c2d, c3d, stream);
  return launch_decode<T, 1, 64>(logits, B, J, D, H, W, hs, c2d, c3d, stream);
}

}  // namespace mtr

extern "C" int mtr_softargmax_decode(const void* logits, int dtype, int layout, int B, int J, int D,
                                     int H, int W, const mtr_head_params* p, float* coords2d,
                                     float* coords3d_rel, mtr_stream_t stream) {
  if (!logits || !p || !coords2d || !coords3d_rel) return MTR_E_NULL;
  if (B < 0 || J <= 0 || D <= 0 || H <= 0 || W <= 0) return MTR_E_SHAPE;
  if (p->proc_side <= 0 || p->stride_test <= 0) return MTR_E_PARAM;
  if (layout != MTR_NCHW) return MTR_E_DTYPE;  // NHWC logits only exist inside the fused head
  if (B == 0) return MTR_OK;
  const mtr::HeadScale hs = mtr::make_head_scale(*p);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case MTR_F32: return mtr::dispatch_decode<float>(logits, B, J, D, H, W, hs, coords2d, coords3d_rel, s);
    case MTR_F16: return mtr::dispatch_decode<__half>(logits, B, J, D, H, W, hs, coords2d, coords3d_rel, s);
    case MTR_BF16: return mtr::dispatch_decode<__hip_bfloat16>(logits, B, J, D, H, W, hs, coords2d, coords3d_rel, s);
    default: return MTR_E_DTYPE;
  }
}
