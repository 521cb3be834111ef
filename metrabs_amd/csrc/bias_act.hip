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

enum Act { kActNone = 0, kActRelu = 1, kActSilu = 2, kActHardswish = 3 };

template <int ACT>
__device__ __forceinline__ float activate(float x) {
  if constexpr (ACT == kActRelu) return fmaxf(x, 0.0f);
  if constexpr (ACT == kActSilu) return x / (1.0f + __expf(-x));  // at::silu: x / (1 + exp(-x))
  if constexpr (ACT == kActHardswish) return x * fminf(fmaxf(x + 3.0f, 0.0f), 6.0f) * (1.0f / 6.0f);
  return x;
}

template <typename T> struct Vec16 { static constexpr int n = 16 / sizeof(T); };

template <typename T, int ACT, bool RES>
__global__ __launch_bounds__(256) void bias_act_kernel(T* __restrict__ y,
                                                       const float* __restrict__ bias,
                                                       const T* __restrict__ residual,
                                                       long long n_vec, int C, int hw_vec) {
  constexpr int VEC = Vec16<T>::n;
  struct alignas(16) Pack { T v[VEC]; };
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / hw_vec) % C);
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

template <typename T, bool RES>
static int launch_bias_act(void* y, const float* bias, const void* residual, int act,
                           long long n_elems, int C, int HW, hipStream_t stream) {
  constexpr int VEC = Vec16<T>::n;
  if (HW % VEC) return MTR_E_SHAPE;
  const long long n_vec = n_elems / VEC;
  long long blocks = (n_vec + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 workgroups per CU
  const dim3 grid((unsigned)blocks), block(256);
  MTR_CLEAR_STALE();
  switch (act) {
    case kActNone: hipLaunchKernelGGL((bias_act_kernel<T, kActNone, RES>), grid, block, 0, stream, (T*)y, bias, (const T*)residual, n_vec, C, HW / VEC); break;
    case kActRelu: hipLaunchKernelGGL((bias_act_kernel<T, kActRelu, RES>), grid, block, 0, stream, (T*)y, bias, (const T*)residual, n_vec, C, HW / VEC); break;
    case kActSilu: hipLaunchKernelGGL((bias_act_kernel<T, kActSilu, RES>), grid, block, 0, stream, (T*)y, bias, (const T*)residual, n_vec, C, HW / VEC); break;
    case kActHardswish: hipLaunchKernelGGL((bias_act_kernel<T, kActHardswish, RES>), grid, block, 0, stream, (T*)y, bias, (const T*)residual, n_vec, C, HW / VEC); break;
    default: return MTR_E_PARAM;
  }
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
