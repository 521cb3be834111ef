// Shared device/host helpers for libmetrabs_hip.so (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <cmath>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <utility>

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

// (u8 / 255) ** 2.2 (person_detector.py:21, multiperson_model.py:197-198), the 256 values of the
// gamma decode: fp32 division like torch, the power in double, rounded once.  Made on the host and
// handed to the kernels by value (1 KiB of kernel arguments): no per-workgroup f64 pow, no global
// state, graph-safe; the pyramid, the sampler (through the table the pyramid stores) and the
// detector pre-processing all read the same 256 floats.
struct GammaLut { float v[256]; };
inline const GammaLut& gamma_lut_host() {
  static const GammaLut lut = [] {
    GammaLut l;
    for (int i = 0; i < 256; ++i) l.v[i] = (float)std::pow((double)((float)i / 255.0f), (double)2.2f);
    return l;
  }();
  return lut;
}


// More than the default 64 KiB of dynamic LDS: the kernel attribute is set ONCE per (kernel, device)
// and remembered (it used to be a host call in front of every launch of the launch-latency-bound
// small-batch paths).  -> MTR_OK or the hipError_t of the failed hipFuncSetAttribute (> 0: the
// ABI's "hipError_t of a failed launch", mtr_strerror names it).
inline int allow_dynamic_lds(const void* kern, size_t lds_bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> granted;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = granted[std::make_pair(kern, dev)];
  if (have >= lds_bytes) return MTR_OK;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return (int)e;
  have = lds_bytes;
  return MTR_OK;
}

// activations of the backbone epilogues (K10 bias_act.hip, K11 depthwise.hip); codes = the `act`
// argument of their entry points
enum Act { kActNone = 0, kActRelu = 1, kActSilu = 2, kActHardswish = 3 };

template <int ACT>
__device__ __forceinline__ float activate(float x) {
  if constexpr (ACT == kActRelu) return fmaxf(x, 0.0f);
  // at::silu: x / (1 + exp(-x)); the quotient as x * v_rcp_f32 (1 ulp) instead of the ten-instruction
  // IEEE division sequence -- K11 is VALU-bound and 2e-7 relative is far inside the backbone's own noise
  if constexpr (ACT == kActSilu) return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
  if constexpr (ACT == kActHardswish) return x * fminf(fmaxf(x + 3.0f, 0.0f), 6.0f) * (1.0f / 6.0f);
  return x;
}

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

// Unsigned 32-bit division by a launch-time constant as a multiply-high and two shifts (Granlund &
// Montgomery; exact for every 32-bit dividend): round 4 -- the grid-stride kernels derived
// (plane, row, column) of a work item with two or three 64-BIT divisions and modulos, ~150 VALU instructions
// each: in the pyramid kernel twice the instructions of the tile's own work.
struct FastDiv {
  unsigned m, s1, s2, d;
};
inline FastDiv make_fastdiv(unsigned d) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;  // ceil(log2 d)
  FastDiv f;
  f.m = (unsigned)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l > 1 ? l - 1 : 0;
  f.d = d;
  return f;
}
__device__ __forceinline__ unsigned fastdiv(unsigned n, const FastDiv& f) {
  const unsigned t = __umulhi(f.m, n);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}


// ---- cross-lane butterflies.  Inside a 16-lane DPP row the partner exchange is a VALU-rate DPP
// move (quad_perm xor-1, xor-2, then row_ror 4 and 8), not an LDS-crossbar ds_bpermute; only the
// two row-crossing steps of a 64-lane reduction go through __shfl_xor.
constexpr int kDppXor1 = 0xB1;    // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;    // quad_perm [2,3,0,1]
constexpr int kDppQuadBcast1 = 0x55, kDppQuadBcast2 = 0xAA, kDppQuadBcast3 = 0xFF;  // quad_perm [k,k,k,k]: lane k of every quad
constexpr int kDppRor4 = 0x124;   // row_ror:4
constexpr int kDppRor8 = 0x128;   // row_ror:8
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror: lane i <-> 7 - i inside each 8 lanes

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// every lane of an aligned group of WIDTH lanes (8, 16, 32 or 64) receives the group's sum / max.
// WIDTH 8: after the two quad steps every lane holds its quad's value, and row_half_mirror pairs
// each lane with one of the other quad of its 8.
template <int WIDTH, typename T>
__device__ __forceinline__ T group_sum(T v) {
  static_assert(WIDTH == 8 || WIDTH == 16 || WIDTH == 32 || WIDTH == 64, "DPP-row fractions / waves");
  v += dpp_move<kDppXor1>(v);
  v += dpp_move<kDppXor2>(v);
  if (WIDTH == 8) return v + dpp_move<kDppHalfMirror>(v);
  v += dpp_move<kDppRor4>(v);
  v += dpp_move<kDppRor8>(v);
  if (WIDTH >= 32) v += __shfl_xor(v, 16, kWave);
  if (WIDTH == 64) v += __shfl_xor(v, 32, kWave);
  return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
  static_assert(WIDTH == 8 || WIDTH == 16 || WIDTH == 32 || WIDTH == 64, "DPP-row fractions / waves");
  v = fmaxf(v, dpp_move<kDppXor1>(v));
  v = fmaxf(v, dpp_move<kDppXor2>(v));
  if (WIDTH == 8) return fmaxf(v, dpp_move<kDppHalfMirror>(v));
  v = fmaxf(v, dpp_move<kDppRor4>(v));
  v = fmaxf(v, dpp_move<kDppRor8>(v));
  if (WIDTH >= 32) v = fmaxf(v, __shfl_xor(v, 16, kWave));
  if (WIDTH == 64) v = fmaxf(v, __shfl_xor(v, 32, kWave));
  return v;
}

// exp(x - m) for x <= m as ONE fma + ONE v_exp_f32: exp2(x*log2e - m*log2e); pass
// neg_m_l2e = -m*log2e.  The argument's rounding error is |x-m|*2^-24 relative, i.e. largest on
// the terms whose softmax weight exp(x-m) is negligible: weighted by the probabilities the error
// of a soft-argmax sum stays at the 1-2 ulp of v_exp_f32 itself.
constexpr float kLog2e = 1.44269504088896340736f;
__device__ __forceinline__ float exp_shifted(float x, float neg_m_l2e) {
  return __builtin_amdgcn_exp2f(fmaf(x, kLog2e, neg_m_l2e));
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

// ---- buffer loads: one wave-uniform descriptor, per-lane byte offset in a VGPR, wave-uniform
// byte offset in an SGPR -> no per-load 64-bit VALU address arithmetic.
using buffer_rsrc_t = decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0));
__device__ __forceinline__ buffer_rsrc_t make_rsrc(const void* wave_uniform_base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wave_uniform_base), (short)0,
                                           (int)bytes, 0x00020000);
}
template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {  // provably wave-uniform for hipcc
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const T*)(((unsigned long long)hi << 32) | lo);
}
// AUX = cache-policy bits of the buffer instruction (0 = default, 2 = nt: streamed once, do not keep)
template <typename T, int VEC, int AUX = 0>
__device__ __forceinline__ void buffer_load_vec(buffer_rsrc_t rsrc, int voff_bytes, int soff_bytes,
                                                float (&out)[VEC]) {
  if constexpr (VEC == 4 && sizeof(T) == 4) {
    // (bind the builtin's own 16-byte vector type with auto: converting it to a differently
    // declared vector type silently splats element 0)
    const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_bytes, soff_bytes, AUX);
    static_assert(sizeof(raw) == 16, "b128 load");
    const float4 f = __builtin_bit_cast(float4, raw);
    out[0] = f.x; out[1] = f.y; out[2] = f.z; out[3] = f.w;
  } else if constexpr (VEC == 8 && sizeof(T) == 2) {
    const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_bytes, soff_bytes, AUX);
    static_assert(sizeof(raw) == 16, "b128 load");
    struct Pack8 { T h[8]; };
    const Pack8 pk = __builtin_bit_cast(Pack8, raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = to_f32(pk.h[i]);
  } else if constexpr (VEC == 4 && sizeof(T) == 2) {
    const auto raw = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff_bytes, soff_bytes, AUX);
    static_assert(sizeof(raw) == 8, "b64 load");
    struct Pack { T h[4]; };
    const Pack pk = __builtin_bit_cast(Pack, raw);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = to_f32(pk.h[i]);
  } else if constexpr (VEC == 1 && sizeof(T) == 4) {
    out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_bytes, soff_bytes, AUX));
  } else {
    static_assert(VEC == 1 && sizeof(T) == 2, "unsupported buffer load shape");
    const unsigned short raw = __builtin_amdgcn_raw_buffer_load_b16(rsrc, voff_bytes, soff_bytes, AUX);
    out[0] = to_f32(__builtin_bit_cast(T, raw));
  }
}

// 1/x in fp64 from v_rcp_f64 + two Newton steps (full double accuracy for normal x > 0): the
// decode epilogue needs two reciprocals per joint instead of ten IEEE divisions.
__device__ __forceinline__ double fast_rcp64(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// exp(x) for x <= 0 in fp64, ~2e-13 relative: t = x*log2(e) = n + f, |f| <= 1/2, 2^f by a degree-11
// Taylor polynomial of exp(f ln2) in Horner form (11 fma), scaled by 2^n through the exponent
// field.  ocml's exp() spends ~4x the instructions on special cases and the last ulp, which a
// softmax weight does not need (its error is relative to a term that is then summed in fp64 and
// rounded to fp32).  Terms below 2^-1000 are flushed to 0.
__device__ __forceinline__ double exp_neg64(double x) {
  const double t = x * 1.4426950408889634;  // log2(e)
  if (!(t > -1000.0)) return 0.0;            // also catches -inf / NaN
  const double n = rint(t);
  const double z = (t - n) * 0.6931471805599453;  // f * ln2, |z| <= 0.347
  double p = 2.505210838544172e-08;               // 1/11!
  p = fma(p, z, 2.755731922398589e-07);
  p = fma(p, z, 2.755731922398589e-06);
  p = fma(p, z, 2.48015873015873e-05);
  p = fma(p, z, 1.984126984126984e-04);
  p = fma(p, z, 1.388888888888889e-03);
  p = fma(p, z, 8.333333333333333e-03);
  p = fma(p, z, 4.166666666666666e-02);
  p = fma(p, z, 1.666666666666667e-01);
  p = fma(p, z, 0.5);
  p = fma(p, z, 1.0);
  p = fma(p, z, 1.0);
  const long long bits = __builtin_bit_cast(long long, p) + ((long long)n << 52);  // p * 2^n, n <= 0
  return __builtin_bit_cast(double, bits);
}

// The same polynomial for a COLD path inside a kernel whose hot loop runs at its register limit:
// `zero` is a run-time 0.0 the caller derives from data that only exists at the call site (x - x of
// a finite value; not foldable without fast-math).  Adding it to every coefficient keeps LLVM from
// materialising the twelve f64 constants once, in front of the hot loop, and holding them in 24
// VGPRs through it (which is what made head_rt_kernel<5, NHWC> spill).
__device__ __forceinline__ double exp_neg64_late(double x, double zero) {
  const double t = x * (1.4426950408889634 + zero);
  if (!(t > -1000.0)) return 0.0;
  const double n = rint(t);
  const double z = (t - n) * (0.6931471805599453 + zero);
  double p = 2.505210838544172e-08 + zero;
  p = fma(p, z, 2.755731922398589e-07 + zero);
  p = fma(p, z, 2.755731922398589e-06 + zero);
  p = fma(p, z, 2.48015873015873e-05 + zero);
  p = fma(p, z, 1.984126984126984e-04 + zero);
  p = fma(p, z, 1.388888888888889e-03 + zero);
  p = fma(p, z, 8.333333333333333e-03 + zero);
  p = fma(p, z, 4.166666666666666e-02 + zero);
  p = fma(p, z, 1.666666666666667e-01 + zero);
  p = fma(p, z, 0.5 + zero);
  p = fma(p, z, 1.0 + zero);
  p = fma(p, z, 1.0 + zero);
  const long long bits = __builtin_bit_cast(long long, p) + ((long long)n << 52);
  return __builtin_bit_cast(double, bits);
}

// Expectation of an axis index -> [0,1]: ptu.decode_heatmap dots with linspace(0,1,n)
// (ptu.py:68-70); ptu.linspace(num==1) is the midpoint 0.5 (ptu.py:83-84).
__device__ __forceinline__ float axis_coord(double weighted_index_sum, double total, int n) {
  if (n <= 1) return 0.5f;
  return (float)(weighted_index_sum / total / (double)(n - 1));
}
// same with precomputed reciprocals: inv_nm1 = 1/(n-1) from the host (AxisInv), or < 0 for n == 1
struct AxisInv { double w, h, d; };
inline AxisInv make_axis_inv(int W, int H, int D) {
  AxisInv a;
  a.w = W > 1 ? 1.0 / (W - 1) : -1.0;
  a.h = H > 1 ? 1.0 / (H - 1) : -1.0;
  a.d = D > 1 ? 1.0 / (D - 1) : -1.0;
  return a;
}
__device__ __forceinline__ float axis_coord_rcp(double weighted_index_sum, double inv_total,
                                                double inv_nm1) {
  if (inv_nm1 < 0.0) return 0.5f;
  return (float)(weighted_index_sum * inv_total * inv_nm1);
}

}  // namespace mtr
