// K10: per-channel bias + activation in place on an NCHW activation tensor.
//
// Not part of the reference's hot path: a fused epilogue for the PyTorch-ROCm
// backbone's inference copy.  With batch norm folded into the convolutions (backbones.fold_batchnorm)
// every conv is followed by "+ bias" and an activation, which PyTorch-ROCm runs as two elementwise
// kernels (MIOpen adds the bias in a separate pass): 4 passes over the activation, 21 % of the
// folded EfficientNetV2-S forward.  This is one pass in place (optionally + the block's skip
// connection, y = act(y + bias) + residual, a third elementwise kernel): HBM-bound, one 16-byte load and one
// 16-byte store per lane, the channel of a vector from one integer division (H*W % VEC == 0, so a
// vector never straddles two channels).
#include "common.h"

namespace mtr {

template <typename T> struct Vec16 { static constexpr int n = 16 / sizeof(T); };

// WIDE: more than 2^32 - 1 vectors (64-bit index arithmetic, the round-1 form)
template <typename T, int ACT, bool RES, bool WIDE>
__global__ __launch_bounds__(256) void bias_act_kernel(T* __restrict__ y,
                                                       const float* __restrict__ bias,
                                                       const T* __restrict__ residual,
                                                       long long n_vec, int C, int hw_vec, FastDiv by_hw,
                                                       FastDiv by_c) {
  constexpr int VEC = Vec16<T>::n;
  struct alignas(16) Pack { T v[VEC]; };
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec;
       i += (long long)gridDim.x * blockDim.x) {
    int c;
    if constexpr (WIDE) {
      c = (int)((i / hw_vec) % C);
    } else {
      const unsigned row = fastdiv((unsigned)i, by_hw);   // (b, c) row of the vector
      c = (int)(row - fastdiv(row, by_c) * by_c.d);
    }
    const float b = bias[c];
    Pack p = *reinterpret_cast<const Pack*>(y + i * VEC);
    Pack q;
    if constexpr (RES) q = *reinterpret_cast<const Pack*>(residual + i * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float r = activate<ACT>(to_f32(p.v[e]) + b);
      if constexpr (RES) r += to_f32(q.v[e]);  // the block's skip connection, added after the activation
      if constexpr (sizeof(T) == 4) p.v[e] = r; else p.v[e] = T(r);
    }
    *reinterpret_cast<Pack*>(y + i * VEC) = p;
  }
}

// Same pass, additionally the mean over H*W of every (b, c) row of the RESULT (as stored): what the
// squeeze-excite block behind a depthwise convolution starts with (x.mean((2, 3))), otherwise one
// more reduction kernel reading the activation again.  LPR lanes share a row (its vectors strided
// over them), 64 / LPR rows per wave; the row sum is a shuffle butterfly inside the LPR lanes.
template <typename T, int ACT, int LPR>
__global__ __launch_bounds__(256) void bias_act_rowmean_kernel(T* __restrict__ y,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ row_mean,
                                                               long long n_rows, int C, int hw_vec,
                                                               float inv_hw) {
  constexpr int VEC = Vec16<T>::n;
  struct alignas(16) Pack { T v[VEC]; };
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long row = wave * (64 / LPR) + lane / LPR;
  const bool live = row < n_rows;
  const long long r = live ? row : n_rows - 1;  // idle lanes recompute the last row, no store
  const float b = bias[(int)(r % C)];
  T* base = y + r * hw_vec * VEC;
  float sum = 0.0f;
  for (int v = sub; v < hw_vec; v += LPR) {
    Pack p = *reinterpret_cast<const Pack*>(base + (long long)v * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float a = activate<ACT>(to_f32(p.v[e]) + b);
      if constexpr (sizeof(T) == 4) p.v[e] = a; else p.v[e] = T(a);
      sum += to_f32(p.v[e]);
    }
    if (live) *reinterpret_cast<Pack*>(base + (long long)v * VEC) = p;
  }
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
  if (live && sub == 0) row_mean[row] = sum * inv_hw;
}

template <typename T, int LPR>
static int launch_bias_act_rowmean(void* y, const float* bias, float* row_mean, int act,
                                   long long n_rows, int C, int HW, hipStream_t stream) {
  constexpr int VEC = Vec16<T>::n;
  const long long waves = (n_rows + 64 / LPR - 1) / (64 / LPR);
  const long long blocks = (waves + 3) / 4;
  if (blocks > 0x7fffffffLL) return MTR_E_SHAPE;
  const dim3 grid((unsigned)blocks), block(256);
  const float inv = 1.0f / (float)HW;
  MTR_CLEAR_STALE();
  switch (act) {
    case kActNone: hipLaunchKernelGGL((bias_act_rowmean_kernel<T, kActNone, LPR>), grid, block, 0, stream, (T*)y, bias, row_mean, n_rows, C, HW / VEC, inv); break;
    case kActRelu: hipLaunchKernelGGL((bias_act_rowmean_kernel<T, kActRelu, LPR>), grid, block, 0, stream, (T*)y, bias, row_mean, n_rows, C, HW / VEC, inv); break;
    case kActSilu: hipLaunchKernelGGL((bias_act_rowmean_kernel<T, kActSilu, LPR>), grid, block, 0, stream, (T*)y, bias, row_mean, n_rows, C, HW / VEC, inv); break;
    case kActHardswish: hipLaunchKernelGGL((bias_act_rowmean_kernel<T, kActHardswish, LPR>), grid, block, 0, stream, (T*)y, bias, row_mean, n_rows, C, HW / VEC, inv); break;
    default: return MTR_E_PARAM;
  }
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

template <typename T>
static int dispatch_bias_act_rowmean(void* y, const float* bias, float* row_mean, int act,
                                     long long n_rows, int C, int HW, hipStream_t stream) {
  constexpr int VEC = Vec16<T>::n;
  if (HW % VEC) return MTR_E_SHAPE;
  if (HW / VEC <= 16) return launch_bias_act_rowmean<T, 16>(y, bias, row_mean, act, n_rows, C, HW, stream);
  return launch_bias_act_rowmean<T, 64>(y, bias, row_mean, act, n_rows, C, HW, stream);
}

template <typename T, bool RES>
static int launch_bias_act(void* y, const float* bias, const void* residual, int act,
                           long long n_elems, int C, int HW, hipStream_t stream) {
  constexpr int VEC = Vec16<T>::n;
  if (HW % VEC) return MTR_E_SHAPE;
  const long long n_vec = n_elems / VEC;
  long long blocks = (n_vec + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 workgroups per CU
  const dim3 grid((unsigned)blocks), block(256);
  const bool wide = n_vec > 0xffffffffLL;
  const FastDiv by_hw = make_fastdiv((unsigned)(HW / VEC)), by_c = make_fastdiv((unsigned)C);
  MTR_CLEAR_STALE();
#define MTR_BIAS_ACT_LAUNCH(ACT)                                                                                 \
  if (wide)                                                                                                      \
    hipLaunchKernelGGL((bias_act_kernel<T, ACT, RES, true>), grid, block, 0, stream, (T*)y, bias,                \
                       (const T*)residual, n_vec, C, HW / VEC, by_hw, by_c);                                    \
  else                                                                                                           \
    hipLaunchKernelGGL((bias_act_kernel<T, ACT, RES, false>), grid, block, 0, stream, (T*)y, bias,               \
                       (const T*)residual, n_vec, C, HW / VEC, by_hw, by_c);
  switch (act) {
    case kActNone: MTR_BIAS_ACT_LAUNCH(kActNone) break;
    case kActRelu: MTR_BIAS_ACT_LAUNCH(kActRelu) break;
    case kActSilu: MTR_BIAS_ACT_LAUNCH(kActSilu) break;
    case kActHardswish: MTR_BIAS_ACT_LAUNCH(kActHardswish) break;
    default: return MTR_E_PARAM;
  }
#undef MTR_BIAS_ACT_LAUNCH
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

}  // namespace mtr

extern "C" int mtr_bias_act_nchw(void* y, int dtype, const float* bias, const void* residual, int act,
                                 long long B, int C, int HW, mtr_stream_t stream) {
  if (!y || !bias) return MTR_E_NULL;
  if (B < 0 || C <= 0 || HW <= 0) return MTR_E_SHAPE;
  if (((uintptr_t)y % 16) || ((uintptr_t)residual % 16)) return MTR_E_ALIGN;
  if (B == 0) return MTR_OK;
  const long long n = B * C * HW;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case MTR_F32:
      return residual ? mtr::launch_bias_act<float, true>(y, bias, residual, act, n, C, HW, s)
                      : mtr::launch_bias_act<float, false>(y, bias, nullptr, act, n, C, HW, s);
    case MTR_F16:
      return residual ? mtr::launch_bias_act<__half, true>(y, bias, residual, act, n, C, HW, s)
                      : mtr::launch_bias_act<__half, false>(y, bias, nullptr, act, n, C, HW, s);
    case MTR_BF16:
      return residual ? mtr::launch_bias_act<__hip_bfloat16, true>(y, bias, residual, act, n, C, HW, s)
                      : mtr::launch_bias_act<__hip_bfloat16, false>(y, bias, nullptr, act, n, C, HW, s);
    default: return MTR_E_DTYPE;
  }
}

extern "C" int mtr_bias_act_rowmean_nchw(void* y, int dtype, const float* bias, int act, long long B,
                                         int C, int HW, float* row_mean, mtr_stream_t stream) {
  if (!y || !bias || !row_mean) return MTR_E_NULL;
  if (B < 0 || C <= 0 || HW <= 0) return MTR_E_SHAPE;
  if ((uintptr_t)y % 16) return MTR_E_ALIGN;
  if (B == 0) return MTR_OK;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case MTR_F32: return mtr::dispatch_bias_act_rowmean<float>(y, bias, row_mean, act, B * C, C, HW, s);
    case MTR_F16: return mtr::dispatch_bias_act_rowmean<__half>(y, bias, row_mean, act, B * C, C, HW, s);
    case MTR_BF16: return mtr::dispatch_bias_act_rowmean<__hip_bfloat16>(y, bias, row_mean, act, B * C, C, HW, s);
    default: return MTR_E_DTYPE;
  }
}
