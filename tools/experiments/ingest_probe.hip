// Per-CU ingest rate probe (developer tool): how fast can ONE workgroup per CU pull an L2-resident
// stream into LDS -- (A) global_load_lds_dwordx4 (LDS-DMA), (B) global_load_dwordx4 + ds_write_b128?
// Build:  hipcc --offload-arch=gfx950 -O3 -o ingest_probe tools/experiments/ingest_probe.hip
// Run:    ./ingest_probe            (prints GB/s per CU for a few wave counts and depths)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// every wave copies JPW KiB per stage; DEPTH stages in flight; stage = waves * JPW KiB
template <int MODE, int JPW, int DEPTH>
__global__ void ingest(const char* __restrict__ src, size_t per_wg, size_t stride, int stages, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), waves = blockDim.x >> 6;
  const char* base = src + (size_t)blockIdx.x * stride;
  {
    const unsigned long long u = (unsigned long long)base;
    base = (const char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)u));
  }
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
  const unsigned stage_bytes = waves * JPW * 1024;
  unsigned voff = (wave * JPW) * 1024 + lane * 16;
  float acc = 0.f;
  if (MODE == 0) {
    for (int s = 0; s < DEPTH - 1; ++s) {
#pragma unroll
      for (int j = 0; j < JPW; ++j) dma16(base, voff + j * 1024, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (s % DEPTH) * stage_bytes + (wave * JPW + j) * 1024)));
      voff += stage_bytes;
    }
    for (int s = 0; s < stages; ++s) {
      wait_vm<(DEPTH - 2) * JPW>();
      __syncthreads();
      acc += reinterpret_cast<const float*>(smem + (s % DEPTH) * stage_bytes)[threadIdx.x];
      const int nx = s + DEPTH - 1;
#pragma unroll
      for (int j = 0; j < JPW; ++j) dma16(base, voff + j * 1024, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (nx % DEPTH) * stage_bytes + (wave * JPW + j) * 1024)));
      voff += stage_bytes;
      if (voff + stage_bytes > per_wg) voff = (wave * JPW) * 1024 + lane * 16;
    }
    wait_vm<0>();
  } else {
    float4 r[JPW];
    for (int s = 0; s < stages; ++s) {
#pragma unroll
      for (int j = 0; j < JPW; ++j) r[j] = *reinterpret_cast<const float4*>(base + voff + j * 1024);
      voff += stage_bytes;
      if (voff + stage_bytes > per_wg) voff = (wave * JPW) * 1024 + lane * 16;
      __syncthreads();
      acc += reinterpret_cast<const float*>(smem + (s & 1) * stage_bytes)[threadIdx.x];
#pragma unroll
      for (int j = 0; j < JPW; ++j)
        *reinterpret_cast<float4*>(smem + ((s + 1) & 1) * stage_bytes + (wave * JPW + j) * 1024 + lane * 16) = r[j];
    }
  }
  if (acc == 12345.f) sink[0] = acc;
}

template <int MODE, int JPW, int DEPTH>
static void run(const char* name, const char* src, size_t per_wg, int threads, int wgs, float* sink, bool shared = false) {
  const int waves = threads / 64, stages = 400;
  const size_t lds = (size_t)DEPTH * waves * JPW * 1024;
  hipFuncSetAttribute((const void*)ingest<MODE, JPW, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    // shared (round 6): every workgroup streams the SAME 640 KB (stride 0) -- all L2 hits after the first touch per XCD:
    // the L2 -> CU ceiling with every CU pulling, which the private-region runs (Infinity Cache bound) cannot show
    hipLaunchKernelGGL((ingest<MODE, JPW, DEPTH>), dim3(wgs), dim3(threads), lds, 0, src, per_wg, shared ? (size_t)0 : per_wg, stages, sink);
    hipEventRecord(b); hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)wgs * stages * waves * JPW * 1024;
  printf("%-34s %s threads %4d wgs %4d  %7.1f us  %7.1f GB/s per WG  %6.2f TB/s chip\n", name, shared ? "SHARED " : "private", threads, wgs, ms * 1e3,
         bytes / wgs / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t per_wg = 640 * 1024;
  char* src; float* sink;
  hipMalloc(&src, per_wg * 512); hipMemset(src, 1, per_wg * 512); hipMalloc(&sink, 16);
  for (int wgs : {256, 512, 64}) {
    run<0, 4, 2>("lds-dma  4 KiB/wave depth 2", src, per_wg, 256, wgs, sink);
    run<0, 4, 4>("lds-dma  4 KiB/wave depth 4", src, per_wg, 256, wgs, sink);
    run<0, 8, 4>("lds-dma  8 KiB/wave depth 4", src, per_wg, 256, wgs, sink);
    run<0, 4, 4>("lds-dma  4 KiB/wave depth 4", src, per_wg, 512, wgs, sink);
    run<1, 4, 2>("registers 4 KiB/wave", src, per_wg, 256, wgs, sink);
    run<1, 8, 2>("registers 8 KiB/wave", src, per_wg, 256, wgs, sink);
    run<1, 4, 2>("registers 4 KiB/wave", src, per_wg, 512, wgs, sink);
  }
  for (int wgs : {256, 512}) {
    run<0, 4, 4>("lds-dma  4 KiB/wave depth 4", src, per_wg, 256, wgs, sink, true);
    run<0, 8, 4>("lds-dma  8 KiB/wave depth 4", src, per_wg, 256, wgs, sink, true);
    run<0, 4, 4>("lds-dma  4 KiB/wave depth 4", src, per_wg, 512, wgs, sink, true);
    run<0, 2, 4>("lds-dma  2 KiB/wave depth 4", src, per_wg, 512, wgs, sink, true);
    run<1, 4, 2>("registers 4 KiB/wave", src, per_wg, 256, wgs, sink, true);
    run<1, 4, 2>("registers 4 KiB/wave", src, per_wg, 512, wgs, sink, true);
  }
  return 0;
}
