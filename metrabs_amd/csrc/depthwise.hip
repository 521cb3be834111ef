// K11: depthwise 3x3 convolution with the K10 epilogue (bias + activation, optionally the per-plane
// mean for the squeeze-excite block behind it) in one pass, NCHW.
//
// Like K10 this sits outside the reference's hot path: it serves the
// PyTorch-ROCm backbone's inference copy (backbones.fold_batchnorm(fused_epilogue=True)).  MIOpen
// runs these layers with its naive direct kernel and PyTorch's own depthwise kernel takes ~61 us
// per layer at the bench shape, followed by the bias / activation / mean passes; here a (b, c)
// plane is read once and written once: HBM-bound, the taps re-read the plane out of L1.
// Mapping: LPP lanes share a plane (its groups of four horizontally adjacent outputs strided over
// them), 64 / LPP planes per wave; each lane keeps the channel's nine weights in registers.
#include "common.h"

namespace mtr {

template <typename T, int ACT, int STRIDE>
__global__ __launch_bounds__(256) void depthwise3x3_kernel(
    const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    T* __restrict__ y, float* __restrict__ row_mean, long long n_planes, int C, int H, int W, int OH,
    int OW, int pad, int lpp, float inv_hw) {
  constexpr int NIN = 3 * STRIDE + 3;  // input columns feeding four adjacent outputs
  struct alignas(4 * sizeof(T)) Out4 { T v[4]; };
  const int lane = threadIdx.x & 63, sub = lane & (lpp - 1);
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long plane = wave * (64 / lpp) + lane / lpp;
  const bool live = plane < n_planes;
  const long long p = live ? plane : n_planes - 1;  // idle lanes recompute the last plane, no store
  const int c = (int)(p % C);
  float wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = w[c * 9 + k];
  const float b = bias[c];
  const T* xp = x + p * (long long)H * W;
  T* yp = y + p * (long long)OH * OW;
  const int ow4 = OW >> 2, nvec = OH * ow4;
  const bool vec_rows = STRIDE == 1 && pad == 1 && (W & 3) == 0;  // (wave-uniform)
  float sum = 0.0f;
  for (int v = sub; v < nvec; v += lpp) {
    const int oy = v / ow4, ox0 = (v - oy * ow4) << 2;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * STRIDE - pad + ky;
      const bool row_ok = iy >= 0 && iy < H;
      const T* row = xp + (row_ok ? iy : 0) * W;
      float in[NIN];
      if (STRIDE == 1 && vec_rows) {
        // stride 1, pad 1, W % 4 == 0: the four centre taps are one aligned vector, the two outer
        // ones single elements (zero at the plane's edge)
        struct alignas(4 * sizeof(T)) In4 { T v[4]; };
        const In4 mid = *reinterpret_cast<const In4*>(row + ox0);
        const T lft = row[ox0 > 0 ? ox0 - 1 : 0], rgt = row[ox0 + 4 < W ? ox0 + 4 : 0];
        in[0] = (row_ok && ox0 > 0) ? to_f32(lft) : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) in[1 + j] = row_ok ? to_f32(mid.v[j]) : 0.0f;
        in[5] = (row_ok && ox0 + 4 < W) ? to_f32(rgt) : 0.0f;
      } else {
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
          const int ix = ox0 * STRIDE - pad + j;
          const bool ok = row_ok && ix >= 0 && ix < W;
          in[j] = ok ? to_f32(row[ok ? ix : 0]) : 0.0f;
        }
      }
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[o] = fmaf(in[o * STRIDE + kx], wk[ky * 3 + kx], acc[o]);
    }
    Out4 out;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const float r = activate<ACT>(acc[o] + b);
      if constexpr (sizeof(T) == 4) out.v[o] = r; else out.v[o] = T(r);
      sum += to_f32(out.v[o]);
    }
    if (live) *reinterpret_cast<Out4*>(yp + oy * OW + ox0) = out;
  }
  if (row_mean) {
    for (int m = lpp >> 1; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
    if (live && sub == 0) row_mean[plane] = sum * inv_hw;
  }
}

template <typename T, int STRIDE>
static int launch_depthwise(const void* x, const float* w, const float* bias, int act, void* y,
                            float* row_mean, long long n_planes, int C, int H, int W, int OH, int OW,
                            int pad, hipStream_t stream) {
  const int nvec = OH * (OW / 4);
  int lpp = 1;
  while (lpp < 64 && lpp < nvec) lpp <<= 1;
  const long long waves = (n_planes + 64 / lpp - 1) / (64 / lpp);
  const long long blocks = (waves + 3) / 4;
  if (blocks > 0x7fffffffLL) return MTR_E_SHAPE;
  const dim3 grid((unsigned)blocks), block(256);
  const float inv = 1.0f / (float)(OH * OW);
  MTR_CLEAR_STALE();
#define MTR_DW_LAUNCH(A)                                                                          \
  hipLaunchKernelGGL((depthwise3x3_kernel<T, A, STRIDE>), grid, block, 0, stream, (const T*)x, w,  \
                     bias, (T*)y, row_mean, n_planes, C, H, W, OH, OW, pad, lpp, inv)
  switch (act) {
    case kActNone: MTR_DW_LAUNCH(kActNone); break;
    case kActRelu: MTR_DW_LAUNCH(kActRelu); break;
    case kActSilu: MTR_DW_LAUNCH(kActSilu); break;
    case kActHardswish: MTR_DW_LAUNCH(kActHardswish); break;
    default: return MTR_E_PARAM;
  }
#undef MTR_DW_LAUNCH
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

template <typename T>
static int dispatch_depthwise(const void* x, const float* w, const float* bias, int act, void* y,
                              float* row_mean, long long n_planes, int C, int H, int W, int OH,
                              int OW, int stride, int pad, hipStream_t stream) {
  if (stride == 1)
    return launch_depthwise<T, 1>(x, w, bias, act, y, row_mean, n_planes, C, H, W, OH, OW, pad, stream);
  return launch_depthwise<T, 2>(x, w, bias, act, y, row_mean, n_planes, C, H, W, OH, OW, pad, stream);
}

}  // namespace mtr

extern "C" int mtr_depthwise3x3_bias_act(const void* x, int dtype, const float* weight,
                                         const float* bias, int act, long long B, int C, int H, int W,
                                         int stride, int pad, void* y, float* row_mean,
                                         mtr_stream_t stream) {
  if (!x || !weight || !bias || !y) return MTR_E_NULL;
  if (B < 0 || C <= 0 || H <= 0 || W <= 0) return MTR_E_SHAPE;
  if ((stride != 1 && stride != 2) || (pad != 0 && pad != 1)) return MTR_E_PARAM;
  const int OH = (H + 2 * pad - 3) / stride + 1, OW = (W + 2 * pad - 3) / stride + 1;
  if (H + 2 * pad < 3 || W + 2 * pad < 3 || OW % 4 != 0) return MTR_E_SHAPE;
  if ((uintptr_t)y % 16) return MTR_E_ALIGN;
  if (B == 0) return MTR_OK;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case MTR_F32: return mtr::dispatch_depthwise<float>(x, weight, bias, act, y, row_mean, B * C, C, H, W, OH, OW, stride, pad, s);
    case MTR_F16: return mtr::dispatch_depthwise<__half>(x, weight, bias, act, y, row_mean, B * C, C, H, W, OH, OW, stride, pad, s);
    case MTR_BF16: return mtr::dispatch_depthwise<__hip_bfloat16>(x, weight, bias, act, y, row_mean, B * C, C, H, W, OH, OW, stride, pad, s);
    default: return MTR_E_DTYPE;
  }
}
