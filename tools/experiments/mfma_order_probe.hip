// Micro-probe: is a chain of v_mfma_f32_16x16x4_f32 the sequential fma chain over k, bit for bit?
// (tools/experiments; not part of the library.)  For random A [16 x K], B [K x 16] (K = 12: three
// MFMAs) it compares every element of D with four scalar formulations:
//   0  t = 0; for k ascending: t = fma(a_k, b_k, t)         (what aten's antialias filter computes)
//   1  the same, k descending inside each group of 4
//   2  unfused: t = t + a_k * b_k (product rounded first)
//   3  per MFMA: ((p0 + p1) + (p2 + p3)) + t with exact products (pairwise)
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/_build/mfma_order_probe tools/experiments/mfma_order_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int K = 12;

__global__ void probe(const float* A, const float* B, float* D, int n) {  // one wave per problem
  const int p = blockIdx.x, l = threadIdx.x;
  const float* a = A + (size_t)p * 16 * K;
  const float* b = B + (size_t)p * K * 16;
  f32x4 acc = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 4) {
    const float av = a[(l % 16) * K + k0 + l / 16], bv = b[(k0 + l / 16) * 16 + l % 16];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[(size_t)p * 256 + (4 * (l / 16) + r) * 16 + l % 16] = acc[r];
}

__global__ void scalar(const float* A, const float* B, float* D, int mode) {
  const int p = blockIdx.x, i = threadIdx.x / 16, j = threadIdx.x % 16;
  const float* a = A + (size_t)p * 16 * K + i * K;
  const float* b = B + (size_t)p * K * 16 + j;
  float t = 0.0f;
  for (int k0 = 0; k0 < K; k0 += 4) {
    if (mode == 0) for (int k = k0; k < k0 + 4; ++k) t = __fmaf_rn(a[k], b[k * 16], t);
    else if (mode == 1) for (int k = k0 + 3; k >= k0; --k) t = __fmaf_rn(a[k], b[k * 16], t);
    else if (mode == 2) for (int k = k0; k < k0 + 4; ++k) t = __fadd_rn(t, __fmul_rn(a[k], b[k * 16]));
    else {
      const double p0 = (double)a[k0] * b[k0 * 16], p1 = (double)a[k0 + 1] * b[(k0 + 1) * 16];
      const double p2 = (double)a[k0 + 2] * b[(k0 + 2) * 16], p3 = (double)a[k0 + 3] * b[(k0 + 3) * 16];
      t = (float)(((p0 + p1) + (p2 + p3)) + (double)t);
    }
  }
  D[(size_t)p * 256 + i * 16 + j] = t;
}

int main() {
  const int n = 4096;
  std::vector<float> A((size_t)n * 16 * K), B((size_t)n * K * 16);
  srand(1);
  for (int set = 0; set < 3; ++set) {
    for (auto& v : A) v = set == 2 ? (float)(rand() % 256) / 255.0f : (float)rand() / RAND_MAX;
    for (auto& v : B) {
      v = (float)rand() / RAND_MAX * 0.3f;
      if (set >= 1 && rand() % 3 == 0) v = 0.0f;  // banded weights: zeros inside the chain
    }
    float *dA, *dB, *dD, *dS;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, (size_t)n * 256 * 4); hipMalloc(&dS, (size_t)n * 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    probe<<<n, 64>>>(dA, dB, dD, n);
    std::vector<float> D((size_t)n * 256), S((size_t)n * 256);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 4; ++mode) {
      scalar<<<n, 256>>>(dA, dB, dS, mode);
      hipMemcpy(S.data(), dS, S.size() * 4, hipMemcpyDeviceToHost);
      size_t same = 0;
      for (size_t i = 0; i < D.size(); ++i) same += memcmp(&D[i], &S[i], 4) == 0;
      printf("set %d mode %d: %zu of %zu identical (%.4f %%)\n", set, mode, same, D.size(), 100.0 * same / D.size());
    }
    hipFree(dA); hipFree(dB); hipFree(dD); hipFree(dS);
  }
  return 0;
}
