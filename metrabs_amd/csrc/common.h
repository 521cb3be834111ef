// Shared device/host helpers for libmetrabs_hip.so (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "../../include/metrabs_hip.h"

namespace mtr {

constexpr int kWave = 64;

// hipGetLastError() is sticky per host thread: an unrelated earlier runtime call of the process
// (e.g. a device probe) may have left an error behind.  Clear it before a launch, check after.
#define MTR_CLEAR_STALE() (void)hipGetLastError()
#define MTR_CHECK_LAUNCH()                       \
  do {                                           \
    hipError_t e_ = hipGetLastError();           \
    if (e_ != hipSuccess) return (int)e_;        \
  } while (0)

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__half v) { return __half2float(v); }
__device__ __forceinline__ float to_f32(__hip_bfloat16 v) { return __bfloat162float(v); }

// VEC consecutive elements -> fp32 registers.  16-byte loads for f32x4, 8-byte for f16x4/bf16x4.
template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T* __restrict__ p, float (&out)[VEC]) {
  if constexpr (VEC == 1) {
    out[0] = to_f32(p[0]);
  } else if constexpr (sizeof(T) == 4) {
    static_assert(VEC == 4, "f32 vectors are float4");
    const float4 v = *reinterpret_cast<const float4*>(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else {
    static_assert(VEC == 4 && sizeof(T) == 2, "16-bit vectors are 4 wide");
    const uint2 raw = *reinterpret_cast<const uint2*>(p);
    const T* h = reinterpret_cast<const T*>(&raw);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = to_f32(h[i]);
  }
}

// xor-butterfly reductions inside an aligned group of WIDTH lanes (WIDTH a power of two <= 64).
template <int WIDTH, typename T>
__device__ __forceinline__ T group_sum(T v) {
#pragma unroll
  for (int m = WIDTH / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int m = WIDTH / 2; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
  return v;
}

// heatmap coordinate in [0,1] -> crop pixels / millimetres with the reference's fp32 op sequence
// (models/util.py:6-33): px = c*last_center (+ s//2) (+ s//2); mm_xy = px*box/proc; mm_z = c*box.
struct HeadScale {
  float last_center;  // (P-1) - ((P-1) % stride)
  float half_stride;  // stride // 2
  int n_half;         // how many times half_stride is added (centered_stride + legacy bug)
  float box_size_mm;
  float proc_side;
};

inline HeadScale make_head_scale(const mtr_head_params& p) {
  HeadScale s;
  const int last = p.proc_side - 1;
  s.last_center = (float)(last - (last % p.stride_test));
  s.half_stride = (float)(p.stride_test / 2);
  s.n_half = (p.centered_stride ? 1 : 0) + (p.legacy_centered_stride_bug ? 1 : 0);
  s.box_size_mm = p.box_size_mm;
  s.proc_side = (float)p.proc_side;
  return s;
}

__device__ __forceinline__ float heatmap_to_px(float c, const HeadScale& s) {
  float v = __fmul_rn(c, s.last_center);
  if (s.n_half >= 1) v = __fadd_rn(v, s.half_stride);
  if (s.n_half >= 2) v = __fadd_rn(v, s.half_stride);
  return v;
}
__device__ __forceinline__ float heatmap_to_mm_xy(float c, const HeadScale& s) {
  return __fdiv_rn(__fmul_rn(heatmap_to_px(c, s), s.box_size_mm), s.proc_side);
}
__device__ __forceinline__ float heatmap_to_mm_z(float c, const HeadScale& s) {
  return __fmul_rn(c, s.box_size_mm);
}

// Expectation of an axis index -> [0,1]: ptu.decode_heatmap dots with linspace(0,1,n)
// (ptu.py:68-70); ptu.linspace(num==1) is the midpoint 0.5 (ptu.py:83-84).
__device__ __forceinline__ float axis_coord(double weighted_index_sum, double total, int n) {
  if (n <= 1) return 0.5f;
  return (float)(weighted_index_sum / total / (double)(n - 1));
}

}  // namespace mtr
