// Can three bf16 terms per f32 operand (6 or 9 bf16 MFMAs per product) stand in for the f32 MFMA of
// the fused head at its parity bar?  One wave computes C[32x32] = A[32xK] . B[Kx32] four ways and
// the host compares with an fp64 evaluation.  Developer probe:
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/bf16x3_probe.hip -o tools/experiments/_build/bf16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ inline void split3(float x, __bf16& b0, __bf16& b1, __bf16& b2) {
  b0 = (__bf16)x;
  const float r1 = x - (float)b0;
  b1 = (__bf16)r1;
  const float r2 = r1 - (float)b1;
  b2 = (__bf16)r2;
}

// mode 0: f32 MFMA 32x32x2, 16-channel chains carried into f64
// mode 1: bf16x3, 6 terms, one f32 accumulator per 64 channels, carried into f64
// mode 2: bf16x3, 6 terms, carried every 16 channels
// mode 3: bf16x3, 9 terms, per 64 channels
// mode 4: bf16x3, 6 terms, small terms in their own accumulator (a0b0 alone), per 64 channels
__global__ void probe(const float* A, const float* B, int K, int mode, double* C) {
  const int l = threadIdx.x, i = l & 31, g = l >> 5;
  double acc[16];
  for (int r = 0; r < 16; ++r) acc[r] = 0;
  if (mode == 0) {
    for (int k0 = 0; k0 < K; k0 += 16) {
      f32x16 p = {0};
      for (int k = k0; k < k0 + 16; k += 2)
        p = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + g], B[(k + g) * 32 + i], p, 0, 0, 0);
      for (int r = 0; r < 16; ++r) acc[r] += (double)p[r];
    }
  } else {
    const int step = (mode == 2) ? 16 : 64;
    for (int k0 = 0; k0 < K; k0 += step) {
      f32x16 p = {0}, q = {0};
      for (int k = k0; k < k0 + step; k += 16) {
        bf16x8 a[3], b[3];
        for (int e = 0; e < 8; ++e) {
          __bf16 t0, t1, t2;
          split3(A[i * K + k + 8 * g + e], t0, t1, t2);
          a[0][e] = t0; a[1][e] = t1; a[2][e] = t2;
          split3(B[(k + 8 * g + e) * 32 + i], t0, t1, t2);
          b[0][e] = t0; b[1][e] = t1; b[2][e] = t2;
        }
        if (mode == 4) {
          q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], q, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], q, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], q, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], q, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], q, 0, 0, 0);
          p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], p, 0, 0, 0);
        } else {
          // small terms first
          if (mode == 3) {
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], p, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], p, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], p, 0, 0, 0);
          }
          p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], p, 0, 0, 0);
          p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], p, 0, 0, 0);
          p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], p, 0, 0, 0);
          p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], p, 0, 0, 0);
          p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], p, 0, 0, 0);
          p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], p, 0, 0, 0);
        }
      }
      for (int r = 0; r < 16; ++r) acc[r] += (double)p[r] + (double)q[r];
    }
  }
  for (int r = 0; r < 16; ++r) C[(8 * (r >> 2) + 4 * g + (r & 3)) * 32 + i] = acc[r];
}

int main() {
  const int K = 1280;
  std::mt19937 rng(7);
  std::normal_distribution<float> n01(0.f, 1.f);
  for (int variant = 0; variant < 3; ++variant) {
    // 0: default-init weights (|w| ~ 0.016) x N(0,1) features; 1: peaked weights x40; 2: positive (post-activation) features
    std::vector<float> A(32 * K), B(K * 32);
    for (auto& v : A) v = n01(rng) * 0.016f * (variant == 1 ? 40.f : 1.f);
    for (auto& v : B) v = variant == 2 ? std::fabs(n01(rng)) * 2.f : n01(rng);
    std::vector<double> truth(32 * 32, 0.0);
    std::vector<float> seq(32 * 32);
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double s = 0; float f = 0;
        for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * (double)B[k * 32 + j]; f = std::fmaf(A[i * K + k], B[k * 32 + j], f); }
        truth[i * 32 + j] = s; seq[i * 32 + j] = f;
      }
    float *dA, *dB; double* dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 32 * 32 * 8);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    double rms = 0; for (double t : truth) rms += t * t; rms = std::sqrt(rms / truth.size());
    double es = 0, em = 0;
    for (int t = 0; t < 1024; ++t) { double e = std::fabs((double)seq[t] - truth[t]); es += e * e; em = std::fmax(em, e); }
    printf("variant %d: |C| rms %.3f; plain f32 fma chain: rms err %.3e max %.3e\n", variant, rms, std::sqrt(es / 1024), em);
    for (int mode = 0; mode < 5; ++mode) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, K, mode, dC);
      std::vector<double> C(32 * 32);
      hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
      es = 0; em = 0;
      for (int t = 0; t < 1024; ++t) { double e = std::fabs(C[t] - truth[t]); es += e * e; em = std::fmax(em, e); }
      printf("  mode %d: rms err %.3e max %.3e\n", mode, std::sqrt(es / 1024), em);
    }
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  return 0;
}
