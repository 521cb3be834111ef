// Micro-probe: operand / result layout of v_mfma_f64_16x16x4_f64 and its accumulation order.
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/_build/mfma_f64_layout_probe tools/experiments/mfma_f64_layout_probe.hip
// A[i][k] = 100 i + k, B[k][j] = (k == kk) * (j + 1) for kk = 0..3: D = A[:, kk] x (j + 1) tells which
// (i, j) a lane's 4 result registers hold; then a chain over K = 64 against the sequential f64 fma chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using f64x4 = __attribute__((ext_vector_type(4))) double;

__global__ void layout(double* out) {  // out[kk][lane][r]
  const int l = threadIdx.x;
  for (int kk = 0; kk < 4; ++kk) {
    const double a = 100.0 * (l % 16) + (l / 16);          // hypothesis: A[i = l % 16][k = l / 16]
    const double b = (l / 16 == kk) ? (double)(l % 16 + 1) : 0.0;  // hypothesis: B[k = l / 16][j = l % 16]
    f64x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(kk * 64 + l) * 4 + r] = c[r];
  }
}

constexpr int K = 64;
__global__ void chain(const double* A, const double* B, double* D) {  // A [16][K], B [K][16]
  const int l = threadIdx.x;
  f64x4 c = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 4)
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l % 16) * K + k0 + l / 16], B[(k0 + l / 16) * 16 + l % 16], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

int main() {
  double* d;
  hipMalloc(&d, 4 * 64 * 4 * 8);
  layout<<<1, 64>>>(d);
  std::vector<double> h(4 * 64 * 4);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  // D[i][j] = (100 i + kk) * (j + 1)  ->  recover (i, j) of every (lane, r)
  bool consistent = true;
  int map_i[64][4], map_j[64][4];
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const double v0 = h[(0 * 64 + l) * 4 + r], v1 = h[(1 * 64 + l) * 4 + r];
      const double jp1 = v1 - v0;  // (j + 1) * (kk difference = 1)
      const int j = (int)(jp1 + 0.5) - 1;
      const int i = (int)(v0 / (jp1 * 100.0) + 0.5);
      map_i[l][r] = i; map_j[l][r] = j;
      for (int kk = 0; kk < 4; ++kk)
        if (h[(kk * 64 + l) * 4 + r] != (100.0 * i + kk) * (j + 1)) consistent = false;
    }
  printf("operand hypothesis A[i=l%%16][k=l/16], B[k=l/16][j=l%%16]: %s\n", consistent ? "consistent" : "NOT consistent");
  for (int l : {0, 1, 15, 16, 17, 32, 48, 63})
    printf("lane %2d: r0 -> (i=%d, j=%d)  r1 -> (i=%d, j=%d)  r2 -> (i=%d, j=%d)  r3 -> (i=%d, j=%d)\n", l, map_i[l][0],
           map_j[l][0], map_i[l][1], map_j[l][1], map_i[l][2], map_j[l][2], map_i[l][3], map_j[l][3]);
  bool rule = true;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) rule &= map_i[l][r] == 4 * (l / 16) + r && map_j[l][r] == l % 16;
  printf("D[i = 4 (l / 16) + r][j = l %% 16]: %s\n", rule ? "yes" : "no");

  std::vector<double> A(16 * K), B(K * 16), D(256);
  srand(3);
  for (auto& v : A) v = (double)(float)((double)rand() / RAND_MAX);
  for (auto& v : B) v = (double)(rand() % 8);
  double *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dD, 256 * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
  chain<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
  int same = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int i = map_i[l][r], j = map_j[l][r];
      double t = 0.0;
      for (int k = 0; k < K; ++k) t = __builtin_fma(A[i * K + k], B[k * 16 + j], t);
      same += memcmp(&t, &D[l * 4 + r], 8) == 0;
    }
  printf("chain of %d MFMAs vs the sequential f64 fma chain over ascending k: %d of 256 identical\n", K / 4, same);
  return 0;
}
