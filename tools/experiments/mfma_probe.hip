// Micro-probe: what limits small-K MFMA chains on gfx950?  (tools/experiments; not part of the library)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/experiments/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f64x4 = __attribute__((ext_vector_type(4))) double;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void probe_f64(const float* in, double* out, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i];
  __syncthreads();
  f64x4 acc[NACC];
  for (int n = 0; n < NACC; ++n) acc[n] = f64x4{0, 0, 0, 0};
  float af = in[threadIdx.x], bf = in[threadIdx.x + 256];
  double ad = af, bd = bf;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // operands resident in registers as doubles
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, bd, acc[n], 0, 0, 0);
    } else if (MODE == 1) {  // one f32->f64 convert per MFMA (B operand) + one per group (A)
      const double a2 = (double)af;
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        const double b2 = (double)(bf + (float)n);
        acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc[n], 0, 0, 0);
      }
      af += 1.0f;
    } else {  // + operands from LDS (1 + NACC ds_read_b32 per group)
      const float a1 = lds[(it * 64 + lane) & 4095];
      const double a2 = (double)a1;
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        const double b2 = (double)lds[(it * 64 + lane + 80 * (n + 1)) & 4095];
        acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc[n], 0, 0, 0);
      }
    }
  }
  double s = 0;
  for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void probe_f32(const float* in, float* out, int iters) {
  f32x4 acc[NACC];
  for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0, 0, 0, 0};
  float af = in[threadIdx.x], bf = in[threadIdx.x + 256];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

using f32x16 = __attribute__((ext_vector_type(16))) float;

// f32 16x16x4 fed from LDS exactly like the head kernel: per k-step 1 A + NACC B ds_read_b32
template <int NACC, bool BATCH>
__global__ __launch_bounds__(256) void probe_f32_lds(const float* in, float* out, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i & 4095];
  __syncthreads();
  f32x4 acc[NACC];
  for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0, 0, 0, 0};
  const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
  for (int it = 0; it < iters; it += 4) {
    float a[4], b[4][NACC];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] = lds[((it + k) * 4 + fk) * 80 % 4096 + fr];
#pragma unroll
      for (int n = 0; n < NACC; ++n) b[k][n] = lds[4096 + (((it + k) * 4 + fk) * 80 + n * 16) % 4000 + fr];
    }
    if (BATCH) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k][n], acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// f32 32x32x2: one 32x32 tile per wave, operands from registers or from LDS (1 A + 1 B read per MFMA)
template <bool LDS>
__global__ __launch_bounds__(256) void probe_f32_32(const float* in, float* out, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i & 4095];
  __syncthreads();
  f32x16 acc = {0};
  const int lane = threadIdx.x & 63;
  float af = in[threadIdx.x], bf = in[threadIdx.x + 256];
  for (int it = 0; it < iters; it += 4) {
    float a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] = LDS ? lds[(((it + k) * 2 + (lane >> 5)) * 80) % 4000 + (lane & 31)] : af;
      b[k] = LDS ? lds[4096 + (((it + k) * 2 + (lane >> 5)) * 80) % 4000 + (lane & 31)] : bf;
    }
    if (LDS) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc, 0, 0, 0);
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
float time_ms(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

int main() {
  float* in; double* outd; float* outf;
  hipMalloc(&in, 8192 * 4); hipMalloc(&outd, 2048 * 256 * 8); hipMalloc(&outf, 2048 * 256 * 4);
  hipMemset(in, 0, 8192 * 4);
  const int iters = 4000;
  for (int blocks : {256, 1024}) {
    const double mf64 = (double)blocks * 4 * iters;  // wave-MFMA count per accumulator
    auto rep = [&](const char* name, float ms, int nacc, double flop_per) {
      const double n = mf64 * nacc;
      const double per_simd = n / 1024.0;
      printf("%-34s blocks=%4d nacc=%d  %.3f ms  %.1f TF  %.1f ns/MFMA/SIMD\n", name, blocks, nacc, ms,
             n * flop_per / ms / 1e9, ms * 1e6 / per_simd);
    };
    rep("f64 regs", time_ms([&] { probe_f64<0, 4><<<blocks, 256>>>(in, outd, iters); }), 4, 2048);
    rep("f64 regs nacc=8", time_ms([&] { probe_f64<0, 8><<<blocks, 256>>>(in, outd, iters); }), 8, 2048);
    rep("f64 + cvt", time_ms([&] { probe_f64<1, 4><<<blocks, 256>>>(in, outd, iters); }), 4, 2048);
    rep("f64 + cvt + lds", time_ms([&] { probe_f64<2, 4><<<blocks, 256>>>(in, outd, iters); }), 4, 2048);
    rep("f32 regs", time_ms([&] { probe_f32<4><<<blocks, 256>>>(in, outf, iters); }), 4, 2048);
    rep("f32 16x16x4 + lds (jit reads)", time_ms([&] { probe_f32_lds<4, false><<<blocks, 256>>>(in, outf, iters); }), 4, 2048);
    rep("f32 16x16x4 + lds (batched)", time_ms([&] { probe_f32_lds<4, true><<<blocks, 256>>>(in, outf, iters); }), 4, 2048);
    rep("f32 32x32x2 regs", time_ms([&] { probe_f32_32<false><<<blocks, 256>>>(in, outf, iters); }), 1, 4096);
    rep("f32 32x32x2 + lds", time_ms([&] { probe_f32_32<true><<<blocks, 256>>>(in, outf, iters); }), 1, 4096);
  }
  return 0;
}
