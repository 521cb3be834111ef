#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int stride_elems) {
  extern __shared__ unsigned short lds[];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  // lane l reads 8 bytes at element offset: row (l % 16) * stride + (l / 16) * 4
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + (l % 16) * stride_elems + (l / 16) * 4));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {64, 16}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, d, stride);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d (lane: 4 values; value = LDS element index; lane l read address row=(l%%16)*stride + (l/16)*4)\n", stride);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l % 2) ? "\n" : "   |   ");
  }
  return 0;
}
