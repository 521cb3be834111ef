// K9: detector pre-processing -- the step in front of the hot path.
//
// Replaces metrabs_pytorch/multiperson/person_detector.py:
//   :15-20,26-29  target size / padding arithmetic (numpy float32)        -> mtr_detector_geometry
//   :21           images = (u8/255)**2.2                                   \
//   :22-24        torchvision resize (bilinear, antialias when shrinking)   |
//   :25           images ** (1/2.2)                                         |-> detector_pre_kernel
//   :30-33        pad to multiples of 32 with 0.5                          /
//   :47-54        scale_boxes: network frame -> image frame                -> detector_boxes_kernel
// The detector network itself (ultralytics YOLOv8) is third party and stays outside.
//
// torchvision's tensor resize is aten's separable upsample_bilinear2d(_aa) (align_corners=False):
// horizontal pass into an f32 intermediate [H_in, W_out], then vertical, each output a short
// weighted sum.  The CPU kernels (what the reference's CPU path runs) were restated and checked
// BIT-EXACT against torch 2.10 (oracle/cpu_ref.py:detector_preprocess; exploration recorded in
// DESIGN.md): weights in float with the double-typed literals of the C++ source rounding where the
// source does, sums as  t = v0*w0; t = fma(v_j, w_j, t)  (antialias) and  fma(v0, w0, v1*w1)
// (plain bilinear).  This kernel mirrors that arithmetic, so the linear-light resize is bit-exact
// and the result differs from the reference only by the two pow roundings (<= 2 ulp).
//
// One workgroup walks output tiles of TY x 64 (persistent grid).  Per tile:
//   1. weights of its 64 columns and TY rows, raw taps in parallel (4 threads per column), the
//      in-order float sum + division per column by one thread each;
//   2. the uint8 region the tile needs (rows y_lo.., bytes x_lo..) is staged into LDS with aligned
//      4-byte loads, many in flight -- the first version read each tap from global memory in a
//      dependent load -> LUT -> fma chain and spent 150 us on 8 x 1080p;
//   3. horizontal pass out of LDS (byte read + gamma LUT + fma, weights in registers, tap loop
//      unrolled to the template bound KT) into an f32 LDS intermediate [rows][64];
//   4. vertical pass, re-gamma, store; pad pixels are written as 0.5 by the same tile walk.
// Bound: HBM read of the uint8 frames (6.2 MB per 1080p frame); in practice the LDS pipe (two LDS
// reads per horizontal tap, ~10 taps per intermediate value at 1080p).
#include <atomic>
#include <cmath>

#include "common.h"
#include "resize_aa.h"

// developer-only timing ablations (tools/experiments/ablate_detector.py); 0 in the product
#ifndef MTR_DET_ABLATE
#define MTR_DET_ABLATE 0
#endif

namespace mtr {

constexpr int kDTX = 64;     // output columns per tile
constexpr int kDTYMax = 8;   // output rows per tile (host picks <= this)
// x ** float32(1 / 2.2) (person_detector.py:25) for x >= 0, in double: x = m 2^e with m in
// [sqrt(1/2), sqrt(2)), log2(m) = 2 / ln 2 * atanh((m - 1) / (m + 1)) to s^13 (|s| <= 0.172), 2^f by its
// series to f^9 (|f| <= 1/2), relative error ~1e-11 before the one rounding to float: the correctly
// rounded power for all but ~6e-6 of the arguments (1 ulp there; 2.4 M arguments checked against
// 80-bit arithmetic).  ~45 double instructions (full rate on gfx950) where the library's powf is ~180
// float ones with its special cases -- the re-gamma was 10 of the kernel's 70 us -- and closer to
// the reference than powf was: torch's CPU pow is correctly rounded for 98.4 % of the arguments,
// the device powf agreed with it on ~90 %.
__device__ __forceinline__ float pow_inv_gamma(float xf) {
  const double x = (double)xf;
  double m = __builtin_amdgcn_frexp_mant(x);  // [1/2, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;
  e = low ? e - 1 : e;
  const double d = m + 1.0;
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  const double s = (m - 1.0) * r, s2 = s * s;
  constexpr double kL = 2.88539008177792681472;  // 2 / ln 2
  double p = kL / 13.0;
  p = __builtin_fma(p, s2, kL / 11.0);
  p = __builtin_fma(p, s2, kL / 9.0);
  p = __builtin_fma(p, s2, kL / 7.0);
  p = __builtin_fma(p, s2, kL / 5.0);
  p = __builtin_fma(p, s2, kL / 3.0);
  p = __builtin_fma(p, s2, kL);
  const double t = __builtin_fma(s, p, (double)e) * (double)(float)(1.0 / 2.2);
  const double n = __builtin_rint(t);
  const double g = (t - n) * 0.69314718055994530942;
  double q = 1.0 / 362880.0;
  q = __builtin_fma(q, g, 1.0 / 40320.0);
  q = __builtin_fma(q, g, 1.0 / 5040.0);
  q = __builtin_fma(q, g, 1.0 / 720.0);
  q = __builtin_fma(q, g, 1.0 / 120.0);
  q = __builtin_fma(q, g, 1.0 / 24.0);
  q = __builtin_fma(q, g, 1.0 / 6.0);
  q = __builtin_fma(q, g, 0.5);
  q = __builtin_fma(q, g, 1.0);
  q = __builtin_fma(q, g, 1.0);
  const float res = (float)__builtin_amdgcn_ldexp(q, (int)n);
  return xf > 0.0f ? res : 0.0f;
}

using DetLut = GammaLut;  // the host-made gamma table (common.h), a kernel argument
// dynamic LDS: stage [rows_cap][pitch] uint8, then temp [rows_cap][64] f32
template <int KT, bool TAIL>
__global__ __launch_bounds__(256) void detector_pre_kernel(
    const uint8_t* __restrict__ src, size_t src_bytes, int planes, AxisGeom gx, AxisGeom gy,
    int pad_top, int pad_left, int out_h, int out_w, int ty_rows, int rows_cap, int pitch,
    float* __restrict__ out, DetLut lut_in) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
  // gamma LUT replicated 16x ([value][16], lane l reads copy l & 15): two lanes of a 32-lane LDS
  // group share a copy, so a random-pixel lookup is ~1.5-way instead of ~3.5-way on one shared table
  __shared__ __attribute__((aligned(16))) float lut[256 * 16];
  __shared__ float wx[KT][kDTX];  // raw taps [tap][column]
  __shared__ float wy[kDTYMax][KT];
  __shared__ float xtotal[kDTX];
  __shared__ int xmin[kDTX], xsize[kDTX], ymin[kDTYMax], ysize[kDTYMax];

  uint8_t* stage = dyn;
  float* temp = reinterpret_cast<float*>(dyn + (size_t)rows_cap * pitch);

  const int tid = threadIdx.x;
  {
    const float v = lut_in.v[tid];
    const float4 v4 = make_float4(v, v, v, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) reinterpret_cast<float4*>(lut + tid * 16)[j] = v4;
  }
  const float* mylut = lut + (tid & 15);
  // A workgroup owns ONE column tile tx for its whole life (grid = tiles_x * G): the column weights
  // are derived once and stay in registers; the loop walks (plane, row tile) pairs.
  const int tiles_x = (out_w + kDTX - 1) / kDTX, tiles_y = (out_h + ty_rows - 1) / ty_rows;
  const int tx = blockIdx.x % tiles_x, lane0 = blockIdx.x / tiles_x, lanes = gridDim.x / tiles_x;
  const int c = tid & (kDTX - 1), rg = tid >> 6;  // column of the tile, row group 0..3

  // ---- column weights (once): raw taps by 4 threads per column, in-order sum by one, division by all
  AxisSpan sp{0, 0, 0.f, 0.f, 0.f};
  {
    const int ox = tx * kDTX + c - pad_left;  // column in the resized image
    if (ox >= 0 && ox < gx.out_size) sp = axis_span(ox, gx);
    sp.isize = min(sp.isize, KT);  // (cannot bind: the host sized KT from the support)
    for (int j = rg; j < sp.isize; j += 4) wx[j][c] = axis_raw_weight(sp, j, gx.aa);
    if (rg == 0) {
      xmin[c] = sp.imin;
      xsize[c] = sp.isize;  // 0 = pad column
    }
  }
  __syncthreads();
  if (tid < kDTX) {
    float total = 0.0f;
    for (int j = 0; j < xsize[tid]; ++j) total = __fadd_rn(total, wx[j][tid]);
    xtotal[tid] = total;
  }
  __syncthreads();
  const int xs = sp.isize, xm = sp.imin;
  float wreg[KT];
  {
    const float total = xtotal[c];
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const float raw = j < xs ? wx[j][c] : 0.0f;
      wreg[j] = (gx.aa && total != 0.0f) ? __fdiv_rn(raw, total) : raw;
    }
  }
  // bytes of the frame rows this column tile reads: xmin and xmin + xsize are monotone in the column
  int x_lo = 0, x_hi = 0;
  {
    const int c0 = max(0, pad_left - tx * kDTX), c1 = min(kDTX - 1, pad_left + gx.out_size - 1 - tx * kDTX);
    if (c1 >= c0) {
      x_lo = xmin[c0];
      x_hi = min(xmin[c1] + (gx.aa ? xsize[c1] : 2), gx.in_size);
    }
  }
  const int vecs = x_hi > x_lo ? (x_hi - x_lo + 15 + 15) / 16 : 0;  // <= pitch / 16 (host sized it)

  for (long long t = lane0; t < (long long)planes * tiles_y; t += lanes) {
    const int tyi = (int)(t % tiles_y), pl = (int)(t / tiles_y);
    __syncthreads();  // previous tile fully consumed
    // ---- 1. row weights of this tile: thread (r, k) evaluates tap k of row r and, redundantly,
    // the row's in-order float sum (<= KT cheap evaluations; saves two barriers per tile)
    for (int e = tid; e < ty_rows * KT; e += 256) {
      const int r = e / KT, k = e - r * KT;
      const int oy = tyi * ty_rows + r - pad_top;
      AxisSpan sy{0, 0, 0.f, 0.f, 0.f};
      if (oy >= 0 && oy < gy.out_size) sy = axis_span(oy, gy);
      sy.isize = min(sy.isize, KT);
      float w = k < sy.isize ? axis_raw_weight(sy, k, gy.aa) : 0.0f;
      if (gy.aa) {
        float total = 0.0f;
        for (int q = 0; q < sy.isize; ++q) total = __fadd_rn(total, axis_raw_weight(sy, q, 1));
        if (total != 0.0f) w = __fdiv_rn(w, total);
      }
      wy[r][k] = w;
      if (k == 0) {
        ymin[r] = sy.imin;
        ysize[r] = sy.isize;
      }
    }
    __syncthreads();
    // rows of the frame this tile reads (ymin, ymin + ysize monotone in the row)
    int y_lo = 0, n_rows = 0;
    {
      const int r0 = max(0, pad_top - tyi * ty_rows);
      const int r1 = min(ty_rows - 1, pad_top + gy.out_size - 1 - tyi * ty_rows);
      if (r1 >= r0) {
        y_lo = ymin[r0];
        n_rows = min(min(ymin[r1] + (gy.aa ? ysize[r1] : 2), gy.in_size) - y_lo, rows_cap);
      }
    }
    // ---- 2. stage rows y_lo.., bytes x_lo..x_hi-1 with aligned 16-byte loads.  Row r of the stage
    // starts at the frame byte a_r = (row start + x_lo) & ~15; its taps sit at offset
    // (row start + x_lo) & 15.
    const size_t plane_off = (size_t)pl * gy.in_size * gx.in_size;
    // 16-byte loads, two per thread in flight before either is stored (a load-store loop with a
    // bounds branch in it serialised one ~1 us global round trip per word).
    const int n_vecs = (MTR_DET_ABLATE & 1) ? 0 : n_rows * vecs;
    for (int e0 = tid; e0 < n_vecs; e0 += 2 * 256) {
      uint4 v[2];
      int dst[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = min(e0 + i * 256, n_vecs - 1);  // clamped duplicates are never stored
        const int r = e / vecs, k = e - r * vecs;
        const size_t row0 = plane_off + (size_t)(y_lo + r) * gx.in_size + x_lo;
        const size_t a = (row0 & ~(size_t)15) + (size_t)k * 16;
        dst[i] = r * pitch + k * 16;
        if (TAIL) {  // tensor size not a multiple of 16: its last vector is assembled from bytes
          if (a + 16 <= src_bytes) {
            v[i] = *reinterpret_cast<const uint4*>(src + a);
          } else {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int q = 0; q < 16; ++q)
              if (a + q < src_bytes) w[q >> 2] |= (uint32_t)src[a + q] << (8 * (q & 3));
            v[i] = make_uint4(w[0], w[1], w[2], w[3]);
          }
        } else {
          v[i] = *reinterpret_cast<const uint4*>(src + a);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (e0 + i * 256 < n_vecs) *reinterpret_cast<uint4*>(stage + dst[i]) = v[i];
    }
    __syncthreads();
    // ---- 3. horizontal pass out of LDS.  The taps of a column are KT consecutive bytes of the
    // staged row: read them as KT/4 + 1 aligned words, realign with v_alignbyte, split with static
    // shifts (one LDS read per 4 taps instead of one per tap), then LUT + fma per tap.
    // RU rows per thread are in flight together: each row is a dependent read -> LUT -> fma chain,
    // and with 3 workgroups per CU there are not enough waves to hide it otherwise.
    if (xs > 0 && !(MTR_DET_ABLATE & 2)) {
      constexpr int NW = (KT + 3) / 4;  // realigned words
      constexpr int RU = KT <= 12 ? 4 : 2;
      for (int rb = rg; rb < n_rows; rb += 4 * RU) {
        uint32_t al[RU][NW];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int r = min(rb + 4 * u, n_rows - 1);  // clamped duplicates are not stored
          const size_t row0 = plane_off + (size_t)(y_lo + r) * gx.in_size + x_lo;
          const int off = (int)(row0 & 15) + (xm - x_lo);  // byte offset of tap 0 in the staged row
          const uint32_t* wrow = reinterpret_cast<const uint32_t*>(stage + (size_t)r * pitch) + (off >> 2);
          uint32_t raw[NW + 1];
#pragma unroll
          for (int i = 0; i <= NW; ++i) raw[i] = wrow[i];
#pragma unroll
          for (int i = 0; i < NW; ++i) al[u][i] = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], off & 3);
        }
        float acc[RU];
        if (gx.aa) {
#pragma unroll
          for (int u = 0; u < RU; ++u) acc[u] = __fmul_rn(mylut[(al[u][0] & 0xff) << 4], wreg[0]);
#pragma unroll
          for (int j = 1; j < KT; ++j)
            if (j < xs) {
#pragma unroll
              for (int u = 0; u < RU; ++u)
                acc[u] = __fmaf_rn(mylut[((al[u][j >> 2] >> (8 * (j & 3))) & 0xff) << 4], wreg[j], acc[u]);
            }
        } else {
          const int j1 = min(xm + 1, gx.in_size - 1) - xm;  // 0 or 1
#pragma unroll
          for (int u = 0; u < RU; ++u) {
            const float v0 = mylut[(al[u][0] & 0xff) << 4], v1 = mylut[((al[u][0] >> (8 * j1)) & 0xff) << 4];
            acc[u] = __fmaf_rn(v0, wreg[0], __fmul_rn(v1, wreg[1]));
          }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u)
          if (rb + 4 * u < n_rows) temp[(rb + 4 * u) * kDTX + c] = acc[u];
      }
    }
    __syncthreads();
    // ---- 4. vertical pass + re-gamma + store (pad pixels: 0.5); a thread's rows (r, r+4) run
    // as independent chains
    {
      constexpr int VU = (kDTYMax + 3) / 4;
      float acc[VU];
      bool live[VU], inside[VU];
#pragma unroll
      for (int u = 0; u < VU; ++u) {
        const int r = rg + 4 * u;
        const int py = tyi * ty_rows + r, px = tx * kDTX + c;
        inside[u] = r < ty_rows && py < out_h && px < out_w;
        const int rr = min(r, ty_rows - 1);
        const int ys = ysize[rr];
        live[u] = inside[u] && ys > 0 && xs > 0 && !(MTR_DET_ABLATE & 4);
        acc[u] = 0.5f;
        if (live[u]) {
          const int y0 = ymin[rr] - y_lo;
          if (gy.aa) {
            float a = __fmul_rn(temp[y0 * kDTX + c], wy[rr][0]);
#pragma unroll
            for (int k = 1; k < KT; ++k)
              if (k < ys) a = __fmaf_rn(temp[(y0 + k) * kDTX + c], wy[rr][k], a);
            acc[u] = a;
          } else {
            const int y1 = min(ymin[rr] + 1, gy.in_size - 1) - y_lo;
            acc[u] = __fmaf_rn(temp[y0 * kDTX + c], wy[rr][0], __fmul_rn(temp[y1 * kDTX + c], wy[rr][1]));
          }
        }
      }
#pragma unroll
      for (int u = 0; u < VU; ++u) {
        if (!inside[u]) continue;
        const int py = tyi * ty_rows + rg + 4 * u, px = tx * kDTX + c;
        out[((size_t)pl * out_h + py) * out_w + px] = live[u] ? pow_inv_gamma(acc[u]) : 0.5f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The streaming form of the same arithmetic (round 3; the default whenever the frame tensor is a
// multiple of 16 bytes).  The tile kernel above runs its four phases one after the other per
// 8-row tile, re-filters the rows two vertically adjacent tiles share (25 % at 1080p) and has
// nothing in flight while it computes: 73 us on 8 x 1080p = 10 % of HBM.  Here
//   * one workgroup of 16 waves per CU owns a 64-column strip of one plane for a run of output rows
//     (a "unit") and walks DOWN the frame: every frame row is filtered horizontally exactly once,
//     into a ring of f32 rows in LDS; an output row is finished (vertical pass, re-gamma, store) as
//     soon as the ring holds its last tap;
//   * wave 0 only loads: whole-wave 1 KiB global -> LDS copies (global_load_lds_dwordx4), 24 per
//     chunk of frame rows (60 rows at 1080p: 4 per computing wave), one chunk (24 KiB) ahead of the
//     15 waves that compute, across unit boundaries; its vmcnt counts nothing but those copies
//     (16 KiB chunks two ahead with counted waits ran the same at 1080p and 12 % slower at 2160p);
//   * one barrier per chunk: the horizontal pass of chunk c and the vertical pass of the rows
//     chunks < c completed run in the same interval (ring >= 2 R + KT rows keeps them apart);
//   * the gamma table is replicated 64x, one copy per lane ([value][64] floats, 64 KiB): every
//     lookup is conflict-free and its LDS address is ONE v_perm_b32 (value byte -> bits 8..15,
//     4 x lane in bits 0..7) instead of extract + shift + add.
// Same weights, same fma order, same table as the tile kernel: the two are bit-identical
// (tests/test_gpu_detector.py).
constexpr int kDSCopies = 24;            // wave copies per chunk: a stage buffer is 24 KiB
constexpr int kDSLa = 1, kDSNbuf = 2;    // chunks in flight / stage buffers
constexpr int kDSGroups = 15;            // computing waves = row groups
constexpr int kDSLutBytes = 256 * 64 * 4;
constexpr int kDSRowsMax = 60;           // frame rows per chunk (4 per computing wave)

struct DetStreamArgs {
  const uint8_t* src;
  size_t src_bytes;
  int planes;
  AxisGeom gx, gy;
  int pad_top, pad_left, out_h, out_w;
  int n_seg, vseg;   // a plane's valid output rows are dealt to n_seg units of vseg rows (the pad rows go
                     // with the first / last one)
  int seg_cap;       // rows of the per-unit row table
  int R;             // frame rows per chunk: R * vecs <= 1024
  int ring;          // rows of the horizontal-pass ring (>= 2 R + KT)
  float* out;
};

// One wave-wide 1 KiB copy global -> LDS: lane L reads 16 bytes at sbase + voff and they land at
// lds_addr + 16 L (sbase, lds_addr wave-uniform).
__device__ __forceinline__ void ds_dma16(const void* sbase, unsigned voff, unsigned lds_addr) {
  const unsigned long long p = (unsigned long long)sbase;
  const unsigned long long su = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)p);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :
               : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(su)
               : "memory", "m0");  // (M0 is overwritten: the register allocator must know)
}
// Workgroup barrier without the fence __syncthreads() carries (s_waitcnt vmcnt(0) would drain the
// loading wave's copies and make every computing wave wait for its output stores): LDS traffic of
// the caller is complete (lgkmcnt), global traffic stays in flight.
__device__ __forceinline__ void ds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void ds_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// rows of the frame a unit reads: [y_lo, y_hi) (empty when it holds pad rows only)
struct DetUnit {
  int pl, o0, o1, y_lo, y_hi, n_chunks;
};
__device__ __forceinline__ DetUnit det_unit(const DetStreamArgs& a, long long t) {
  DetUnit u;
  const int seg = (int)(t % a.n_seg);
  u.pl = (int)(t / a.n_seg);
  u.o0 = seg == 0 ? 0 : min(a.pad_top + seg * a.vseg, a.out_h);
  u.o1 = seg == a.n_seg - 1 ? a.out_h : min(a.pad_top + (seg + 1) * a.vseg, a.out_h);
  const int r0 = max(u.o0, a.pad_top), r1 = min(u.o1, a.pad_top + a.gy.out_size) - 1;
  u.y_lo = u.y_hi = 0;
  if (r1 >= r0) {  // (imin and imin + isize are monotone in the row)
    const AxisSpan s0 = axis_span(r0 - a.pad_top, a.gy), s1 = axis_span(r1 - a.pad_top, a.gy);
    u.y_lo = s0.imin;
    u.y_hi = min(s1.imin + (a.gy.aa ? s1.isize : 2), a.gy.in_size);
  }
  u.n_chunks = (u.y_hi - u.y_lo + a.R - 1) / a.R;
  return u;
}

template <int KT>
__global__ __launch_bounds__(1024) void detector_stream_kernel(DetStreamArgs a, DetLut lut_in) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
  // (no static LDS: the table sits at LDS address 0, so a lookup's address IS the v_perm result)
  float* lut = reinterpret_cast<float*>(dyn);                       // [256][64]
  uint8_t* stage = dyn + kDSLutBytes;                               // [kDSNbuf][16 KiB]
  float* temp = reinterpret_cast<float*>(stage + kDSNbuf * kDSCopies * 1024);  // [ring][64]
  float* wy = temp + (size_t)a.ring * kDTX;                         // [seg_cap][KT]
  int4* yinfo = reinterpret_cast<int4*>(wy + (size_t)a.seg_cap * KT);  // [seg_cap] (ring slot of tap 0,
                                                                    // taps (0 = pad row), last row + 1, step to tap 1)
  float (*wx)[kDTX] = reinterpret_cast<float (*)[kDTX]>(yinfo + a.seg_cap);  // [KT][64] raw column taps
  float* xtotal = &wx[KT][0];
  int* xmin = reinterpret_cast<int*>(xtotal + kDTX);
  int* xsize = xmin + kDTX;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    const float v = lut_in.v[tid >> 2];
    const float4 v4 = make_float4(v, v, v, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) reinterpret_cast<float4*>(lut + (tid >> 2) * 64 + (tid & 3) * 16)[j] = v4;
  }
  const int tiles_x = (a.out_w + kDTX - 1) / kDTX;
  const int tx = blockIdx.x % tiles_x, lane0 = blockIdx.x / tiles_x, lanes = gridDim.x / tiles_x;
  const long long n_units = (long long)a.planes * a.n_seg;
  const int c = lane;
  const AxisGeom gx = a.gx, gy = a.gy;

  // ---- column weights, once (as the tile kernel: raw taps in parallel, in-order sum by one thread)
  AxisSpan sp{0, 0, 0.f, 0.f, 0.f};
  {
    const int ox = tx * kDTX + c - a.pad_left;
    if (ox >= 0 && ox < gx.out_size) sp = axis_span(ox, gx);
    sp.isize = min(sp.isize, KT);
    for (int j = wave; j < sp.isize; j += 16) wx[j][c] = axis_raw_weight(sp, j, gx.aa);
    if (wave == 0) {
      xmin[c] = sp.imin;
      xsize[c] = sp.isize;
    }
  }
  __syncthreads();
  if (tid < kDTX) {
    float total = 0.0f;
    for (int j = 0; j < xsize[tid]; ++j) total = __fadd_rn(total, wx[j][tid]);
    xtotal[tid] = total;
  }
  __syncthreads();
  const int xs = sp.isize, xm = sp.imin;
  int x_lo = 0, x_hi = 0;
  {
    const int c0 = max(0, a.pad_left - tx * kDTX), c1 = min(kDTX - 1, a.pad_left + gx.out_size - 1 - tx * kDTX);
    if (c1 >= c0) {
      x_lo = xmin[c0];
      x_hi = min(xmin[c1] + (gx.aa ? xsize[c1] : 2), gx.in_size);
    }
  }
  const int vecs = x_hi > x_lo ? (x_hi - x_lo + 15 + 15) / 16 : 1;  // R * vecs <= 1024 (host)
  const int pitch = vecs * 16;
  int xs_max = sp.isize;  // (uniform) most taps of any column of the strip: a wave holds all 64 columns
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) xs_max = max(xs_max, __shfl_xor(xs_max, o));
  xs_max = __builtin_amdgcn_readfirstlane(xs_max);

  if (wave == 0) {
    // ================= the loading wave =================
    // copy i of a chunk carries vectors e = 64 i + lane = (row e / vecs, vector e % vecs) of the chunk
    unsigned rk[kDSCopies];
#pragma unroll
    for (int i = 0; i < kDSCopies; ++i) {
      const int e = i * 64 + lane, r = e / vecs;
      rk[i] = ((unsigned)r << 16) | (unsigned)(e - r * vecs);
    }
    const unsigned stage0 = (unsigned)(size_t)(const __attribute__((address_space(3))) uint8_t*)stage;
    long long it = lane0;          // unit of the next chunk to issue
    DetUnit iu{0, 0, 0, 0, 0, 0};
    int ic = 0;                    // its chunk
    bool have = false;
    int issued = 0, landed = 0;    // chunks issued / handed over so far
    auto next_chunk = [&]() -> bool {  // positions (iu, ic) on the next chunk to issue
      if (have && ic + 1 < iu.n_chunks) { ++ic; return true; }
      if (have) it += lanes;
      for (; it < n_units; it += lanes) {
        iu = det_unit(a, it);
        if (iu.n_chunks > 0) { ic = 0; have = true; return true; }
      }
      have = false;
      return false;
    };
    auto issue = [&]() {
      if (!next_chunk()) return;
      const int y_c = iu.y_lo + ic * a.R, n_rows = min(a.R, iu.y_hi - y_c);
      const size_t base = ((size_t)iu.pl * gy.in_size + y_c) * gx.in_size + x_lo;
      const unsigned lo4 = (unsigned)(base & 15);
      const uint8_t* sbase = a.src + (base - lo4);
      const size_t limit = a.src_bytes - 16 - (base - lo4);  // (the tensor is a multiple of 16 bytes)
      const unsigned buf = stage0 + (unsigned)(issued % kDSNbuf) * (kDSCopies * 1024);
#pragma unroll
      for (int i = 0; i < kDSCopies; ++i) {
        const int r = min((int)(rk[i] >> 16), n_rows - 1);  // (rows past the chunk: duplicates, never read)
        const size_t off = (size_t)((lo4 + (unsigned)r * (unsigned)gx.in_size) & ~15u) + (rk[i] & 0xffffu) * 16u;
        if (!(MTR_DET_ABLATE & 1)) ds_dma16(sbase, (unsigned)min(off, limit), buf + i * 1024);
      }
      ++issued;
    };
#pragma unroll
    for (int i = 0; i < kDSLa; ++i) issue();
    for (long long t = lane0; t < n_units; t += lanes) {
      const DetUnit u = det_unit(a, t);
      ds_barrier();  // A
      for (int ch = 0; ch < u.n_chunks; ++ch) {
        if (issued - landed - 1 >= kDSLa - 1) ds_wait_vmcnt<(kDSLa - 1) * kDSCopies>();
        else ds_wait_vmcnt<0>();
        ++landed;
        ds_barrier();  // B: chunk `landed - 1` is in LDS, the buffer of chunk `landed - 2` is free
        issue();
      }
      ds_barrier();  // C
      ds_barrier();  // D
    }
    return;
  }

  // ================= the 15 computing waves =================
  const int rg = wave - 1;
  // normalised column weights: in registers (2 / 12 taps) or, for the 24- and 40-tap instantiations, in LDS
  // in place of the raw taps (24 / 40 registers on top of the vertical pass and the double-precision pow spilled)
  constexpr bool kWeightsInLds = KT > 12;
  constexpr int kWregs = kWeightsInLds ? 1 : KT;
  float wreg[kWregs];
  float* wnorm = &wx[0][0];  // [KT][64]: the raw taps, normalised in place by ONE wave (nobody else reads them any
                             // more; visible to the others behind the unit's first barrier)
  {
    const float total = xtotal[c];
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      if constexpr (kWeightsInLds) {
        if (wave == 1) {
          const float raw = j < xs ? wx[j][c] : 0.0f;
          wx[j][c] = (gx.aa && total != 0.0f) ? __fdiv_rn(raw, total) : raw;
        }
      } else {
        const float raw = j < xs ? wx[j][c] : 0.0f;
        wreg[j] = (gx.aa && total != 0.0f) ? __fdiv_rn(raw, total) : raw;
      }
    }
  }
#define DS_W(j) (kWeightsInLds ? wnorm[(j) * kDTX + c] : wreg[kWeightsInLds ? 0 : (j)])
  const unsigned lane4 = (unsigned)lane * 4u;
  // gamma value of byte K (0..3) of word w: LDS address = byte << 8 | 4 lane, one v_perm_b32.  The
  // table is at LDS address 0 (this kernel has no static LDS; checked once below), so the permuted
  // word is used as the address as it is -- through the generic `lut + ...` the compiler keeps an
  // add of the (link-time) base in front of every read.
  if ((unsigned)(size_t)(const __attribute__((address_space(3))) float*)lut != 0u) __builtin_trap();
  using lds_float_ptr = const __attribute__((address_space(3))) float*;
#define DS_LUT(w, K) \
  (*(lds_float_ptr)(size_t)__builtin_amdgcn_perm((w), lane4, 0x0c0c0000u | ((4u + (K)) << 8)))
  const int ctid = tid - 64;  // 0..959
  int job = 0;                // chunks consumed so far by this workgroup (stage buffer = job % 3)
  const int px = tx * kDTX + c;

  for (long long t = lane0; t < n_units; t += lanes) {
    const DetUnit u = det_unit(a, t);
    // ---- the unit's row table: a group of GS lanes per row, lane k evaluates tap k; the row's
    // in-order float sum runs over the group's values (taps past the row's count are 0: x + 0 = x)
    const int n_out = u.o1 - u.o0;
    constexpr int GS = KT <= 16 ? 16 : KT <= 32 ? 32 : 64;
    for (int r = ctid / GS; r < n_out; r += kDSGroups * 64 / GS) {
      const int k = ctid & (GS - 1);
      const int oy = u.o0 + r - a.pad_top;
      AxisSpan sy{0, 0, 0.f, 0.f, 0.f};
      const bool valid = oy >= 0 && oy < gy.out_size;
      if (valid) sy = axis_span(oy, gy);
      sy.isize = min(sy.isize, KT);
      float w = k < sy.isize ? axis_raw_weight(sy, k, gy.aa) : 0.0f;
      if (gy.aa) {
        float total = 0.0f;
#pragma unroll
        for (int q = 0; q < KT; ++q) total = __fadd_rn(total, __shfl(w, q, GS));
        if (total != 0.0f) w = __fdiv_rn(w, total);
      }
      if (k < KT) wy[r * KT + k] = w;
      if (k == 0) {
        int4 inf = make_int4(0, 0, 0, 0);
        if (valid) {
          inf.x = (sy.imin - u.y_lo) % a.ring;
          inf.y = sy.isize;
          inf.z = min(sy.imin + (gy.aa ? sy.isize : 2), gy.in_size);
          inf.w = min(sy.imin + 1, gy.in_size - 1) - sy.imin;
        }
        yinfo[r] = inf;
      }
    }
    ds_barrier();  // A
    int o_done = u.o0;
    const size_t plane_off = (size_t)u.pl * gy.in_size * gx.in_size;

    // vertical pass + re-gamma + store of the output rows whose taps are all in the ring
    auto finish_rows = [&](int rows_done) {
      int o_next = o_done;
      for (;;) {  // (uniform) rows o_done.. whose last tap is in: 64 rows per test
        const int o = o_next + lane;
        const unsigned long long in = __ballot(o < u.o1 && yinfo[min(o, u.o1 - 1) - u.o0].z <= rows_done);
        const int n = in == ~0ull ? 64 : __builtin_ctzll(~in);
        o_next += n;
        if (n < 64) break;
      }
      if (px < a.out_w) {
        for (int o = o_done + rg; o < o_next; o += kDSGroups) {
          const int rr = o - u.o0;
          const int4 inf = yinfo[rr];
          const int ys = __builtin_amdgcn_readfirstlane(inf.y);  // (a wave works on one output row)
          float res = 0.5f;
          if (ys > 0 && xs > 0 && !(MTR_DET_ABLATE & 4)) {
            float acc;
            int slot = inf.x;
            if (gy.aa) {
              const int slot0 = __builtin_amdgcn_readfirstlane(slot);
              if (slot0 + KT <= a.ring) {  // (uniform) the window does not wrap: taps at immediate offsets
                const float* tp = temp + slot0 * kDTX + c;
                const float* wp = wy + rr * KT;
                acc = __fmul_rn(tp[0], wp[0]);
#pragma unroll
                for (int k = 1; k < KT; ++k)
                  if (k < ys) acc = __fmaf_rn(tp[k * kDTX], wp[k], acc);
              } else {
                acc = __fmul_rn(temp[slot * kDTX + c], wy[rr * KT]);
#pragma unroll
                for (int k = 1; k < KT; ++k)
                  if (k < ys) {
                    slot = slot + 1 == a.ring ? 0 : slot + 1;
                    acc = __fmaf_rn(temp[slot * kDTX + c], wy[rr * KT + k], acc);
                  }
              }
            } else {
              int s1 = slot + inf.w;
              s1 = s1 >= a.ring ? s1 - a.ring : s1;
              acc = __fmaf_rn(temp[slot * kDTX + c], wy[rr * KT], __fmul_rn(temp[s1 * kDTX + c], wy[rr * KT + 1]));
            }
            res = (MTR_DET_ABLATE & 8) ? acc : pow_inv_gamma(acc);
          }
          a.out[((size_t)u.pl * a.out_h + o) * a.out_w + px] = res;
        }
      }
      o_done = o_next;
    };

    int slot_c = 0;  // ring slot of the chunk's first row
    for (int ch = 0; ch < u.n_chunks; ++ch) {
      ds_barrier();  // B
      const int y_c = u.y_lo + ch * a.R, n_rows = min(a.R, u.y_hi - y_c);
      const uint8_t* sbuf = stage + (job % kDSNbuf) * (kDSCopies * 1024);
      ++job;
      // ---- horizontal pass: rows rg, rg + 15, ... of the chunk, RU of them in flight per thread
      if (xs > 0 && !(MTR_DET_ABLATE & 2)) {
        constexpr int NW = (KT + 3) / 4;
        constexpr int RU = KT <= 12 ? 4 : KT <= 24 ? 2 : 1;
        // taps every shape of this instantiation has (2 scale + 1 > the next smaller instantiation's bound
        // minus 3): no test in front of them, so they schedule as one block
        constexpr int kTapsSure = KT == 12 ? 4 : KT == 24 ? 8 : KT == kDTaps ? 20 : KT;
        const size_t base = plane_off + (size_t)y_c * gx.in_size + x_lo;
        const unsigned lo4 = (unsigned)(base & 15), w15 = (unsigned)gx.in_size & 15u;
        for (int rb = rg; rb < n_rows; rb += kDSGroups * RU) {
          uint32_t al[RU][NW];
#pragma unroll
          for (int uu = 0; uu < RU; ++uu) {
            const int r = min(rb + kDSGroups * uu, n_rows - 1);
            // byte offset of tap 0 in the staged row: the row starts at the frame byte (base + r W) & ~15
            // (r < 64, pitch < 2^16: 24-bit multiplies, full rate; v_mul_lo_u32 is a quarter-rate op)
            const int off = (int)((lo4 + __umul24((unsigned)r, w15)) & 15u) + (xm - x_lo);
            const uint32_t* wrow =
                reinterpret_cast<const uint32_t*>(sbuf + __umul24((unsigned)r, (unsigned)pitch)) + (off >> 2);
            uint32_t raw[NW + 1];
#pragma unroll
            for (int i = 0; i <= NW; ++i) raw[i] = wrow[i];
#pragma unroll
            for (int i = 0; i < NW; ++i) al[uu][i] = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], off & 3);
          }
          float acc[RU];
          if (gx.aa) {
            // taps past a column's own count carry weight 0: fma(v, 0, acc) = acc exactly (v finite), so
            // the taps run without a per-lane branch up to the strip's largest count (a uniform test)
#pragma unroll
            for (int uu = 0; uu < RU; ++uu) acc[uu] = __fmul_rn(DS_LUT(al[uu][0], 0u), DS_W(0));
#pragma unroll
            for (int j = 1; j < KT; ++j)
              if (j < kTapsSure || j < xs_max) {
#pragma unroll
                for (int uu = 0; uu < RU; ++uu)
                  acc[uu] = __fmaf_rn(DS_LUT(al[uu][j >> 2], (unsigned)(j & 3)), DS_W(j), acc[uu]);
              }
          } else {
            const int j1 = min(xm + 1, gx.in_size - 1) - xm;  // 0 or 1
#pragma unroll
            for (int uu = 0; uu < RU; ++uu) {
              const float v0 = DS_LUT(al[uu][0], 0u);
              const float v1 = j1 ? DS_LUT(al[uu][0], 1u) : v0;
              acc[uu] = __fmaf_rn(v0, DS_W(0), __fmul_rn(v1, DS_W(1)));
            }
          }
#pragma unroll
          for (int uu = 0; uu < RU; ++uu) {
            const int r = rb + kDSGroups * uu;
            if (r < n_rows) {
              int slot = slot_c + r;
              slot = slot >= a.ring ? slot - a.ring : slot;
              temp[slot * kDTX + c] = acc[uu];
            }
          }
        }
      }
      // ---- rows the chunks before this one completed
      finish_rows(y_c);
      slot_c += n_rows;
      slot_c = slot_c >= a.ring ? slot_c - a.ring : slot_c;
    }
    ds_barrier();  // C
    finish_rows(u.y_hi);
    ds_barrier();  // D: the row table is free
  }
#undef DS_LUT
#undef DS_W
}

// person_detector.py:47-54.  in: [n,5] (x1, y1, x2, y2, conf) in the padded network frame;
// out: [n,5] (x, y, w, h, conf) in the image frame.  (w, h as ultralytics' xywh: x2-x1, y2-y1.)
__global__ void detector_boxes_kernel(const float* __restrict__ in, int n, float half_pad_w,
                                      float half_pad_h, float x_factor, float y_factor,
                                      float* __restrict__ outb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x1 = in[i * 5], y1 = in[i * 5 + 1], x2 = in[i * 5 + 2], y2 = in[i * 5 + 3];
  outb[i * 5 + 0] = __fmul_rn(__fsub_rn(x1, half_pad_w), x_factor);
  outb[i * 5 + 1] = __fmul_rn(__fsub_rn(y1, half_pad_h), y_factor);
  outb[i * 5 + 2] = __fmul_rn(__fsub_rn(x2, x1), x_factor);
  outb[i * 5 + 3] = __fmul_rn(__fsub_rn(y2, y1), y_factor);
  outb[i * 5 + 4] = in[i * 5 + 4];
}

}  // namespace mtr

// host-only: the size arithmetic of person_detector.py:15-20,26-29 in float32, as numpy does it
extern "C" int mtr_detector_geometry(int H, int W, int input_size, mtr_detector_geom* g) {
  if (!g) return MTR_E_NULL;
  if (H <= 0 || W <= 0 || input_size <= 0) return MTR_E_SHAPE;
  const float h = (float)H, w = (float)W;
  const float max_side = h > w ? h : w;
  const volatile float factor = (float)input_size / max_side;  // (volatile: no fused re-association)
  const volatile float fw = factor * w, fh = factor * h;
  g->target_w = (int32_t)fw;
  g->target_h = (int32_t)fh;
  if (g->target_w <= 0 || g->target_h <= 0) return MTR_E_SHAPE;
  g->antialias = factor < 1.0f;
  const int pad_h = ((-g->target_h) % 32 + 32) % 32, pad_w = ((-g->target_w) % 32 + 32) % 32;
  g->pad_top = pad_h / 2;
  g->pad_left = pad_w / 2;
  g->out_h = g->target_h + pad_h;
  g->out_w = g->target_w + pad_w;
  g->x_factor = w / (float)g->target_w;
  g->y_factor = h / (float)g->target_h;
  return MTR_OK;
}

// compute units of the current device (queried once per device and process; read-only)
static int device_cu_count() {
  static std::atomic<int> cache[64];  // (zero-initialised; racing first calls store the same value)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// The streaming kernel addresses its gamma table as "LDS address = permuted word": the dynamic LDS has to
// start at address 0, i.e. the kernel must own no static LDS.  Asked of the code object once per
// instantiation (1 = yes, -1 = no: the caller takes the tile kernel); the device-side trap stays as a backstop.
template <typename Kern>
static bool stream_kernel_lds_starts_at_zero(Kern kern) {
  static std::atomic<int> known{0};
  int k = known.load(std::memory_order_relaxed);
  if (k == 0) {
    hipFuncAttributes attr;
    k = (hipFuncGetAttributes(&attr, (const void*)kern) == hipSuccess && attr.sharedSizeBytes == 0) ? 1 : -1;
    if (k < 0) (void)hipGetLastError();
    known.store(k, std::memory_order_relaxed);
  }
  return k > 0;
}

static const mtr::DetLut& detector_lut() { return mtr::gamma_lut_host(); }

template <int KT, bool TAIL>
static int launch_detector_pre_t(const uint8_t* images_u8, int N, int H, int W, const mtr_detector_geom* g,
                               int ty, int rows_cap, int pitch, float* out, hipStream_t stream) {
  const mtr::AxisGeom gx{W, g->target_w, g->antialias}, gy{H, g->target_h, g->antialias};
  const size_t lds = (size_t)rows_cap * pitch + (size_t)rows_cap * mtr::kDTX * sizeof(float);
  auto kern = mtr::detector_pre_kernel<KT, TAIL>;
  if (lds > 48 * 1024) {
    const int rc = mtr::allow_dynamic_lds((const void*)kern, lds);
    if (rc != MTR_OK) return rc;
  }
  // persistent: every workgroup keeps one column tile; G workgroups share its (plane, row tile)
  // pairs.  The grid is sized to what is RESIDENT at once (a workgroup that has to wait for a slot
  // makes a second round: 128 -> 80 us on 8 x 1080p).
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 256, lds) != hipSuccess ||
      per_cu < 1)
    per_cu = 1;
  const int n_cu = device_cu_count();
  const int tiles_x = (g->out_w + mtr::kDTX - 1) / mtr::kDTX;
  const long long pairs = (long long)N * 3 * ((g->out_h + ty - 1) / ty);
  long long G = (long long)per_cu * n_cu / tiles_x;
  if (G < 1) G = 1;
  if (G > pairs) G = pairs;
  const int grid = (int)(G * tiles_x);
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, images_u8, (size_t)N * 3 * H * W, N * 3,
                     gx, gy, g->pad_top, g->pad_left, g->out_h, g->out_w, ty, rows_cap, pitch, out, detector_lut());
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

constexpr int kDetNoStream = -1000;  // (internal: the caller falls back to the tile kernel)
// The streaming kernel: one workgroup per CU.  kDetNoStream when the tensor is not a multiple of
// 16 bytes (its loads are whole 16-byte vectors) or the shape's LDS does not fit.
template <int KT>
static int launch_detector_stream(const uint8_t* images_u8, int N, int H, int W, const mtr_detector_geom* g,
                                  int pitch_cap, float* out, hipStream_t stream) {
  const size_t src_bytes = (size_t)N * 3 * H * W;
  if (src_bytes % 16 || src_bytes < 16) return kDetNoStream;
  mtr::DetStreamArgs a;
  a.src = images_u8;
  a.src_bytes = src_bytes;
  a.planes = N * 3;
  a.gx = mtr::AxisGeom{W, g->target_w, g->antialias};
  a.gy = mtr::AxisGeom{H, g->target_h, g->antialias};
  a.pad_top = g->pad_top;
  a.pad_left = g->pad_left;
  a.out_h = g->out_h;
  a.out_w = g->out_w;
  a.out = out;
  const int vecs_cap = pitch_cap / 16;  // >= the kernel's vectors per staged row
  a.R = mtr::kDSCopies * 64 / vecs_cap;
  if (a.R > mtr::kDSRowsMax) a.R = mtr::kDSRowsMax;
  if (a.R < 1) return kDetNoStream;
  a.ring = 2 * a.R + KT;
  const int n_cu = device_cu_count();
  const int tiles_x = (g->out_w + mtr::kDTX - 1) / mtr::kDTX;
  const int G = n_cu / tiles_x > 0 ? n_cu / tiles_x : 1;  // workgroups per column strip
  // units per strip: two rounds of its G workgroups, at least 8 output rows each, the row table <= 8 KiB
  const int valid = g->target_h;
  int n_seg = (2 * G + a.planes - 1) / a.planes;
  if (n_seg > (valid + 7) / 8) n_seg = (valid + 7) / 8;
  if (n_seg < 1) n_seg = 1;
  const int table_rows = 8192 / (KT * 4 + 16);
  for (;; ++n_seg) {
    a.vseg = (valid + n_seg - 1) / n_seg;
    a.n_seg = (valid + a.vseg - 1) / a.vseg;
    const int pad_bottom = g->out_h - g->pad_top - valid;
    a.seg_cap = a.vseg + (g->pad_top > pad_bottom ? g->pad_top : pad_bottom);
    if (a.n_seg == 1) a.seg_cap = g->out_h;
    if (a.seg_cap <= table_rows || a.vseg == 1) break;
  }
  if (a.seg_cap > table_rows) return kDetNoStream;
  const size_t lds = (size_t)mtr::kDSLutBytes + (size_t)mtr::kDSNbuf * mtr::kDSCopies * 1024 +
                     (size_t)a.ring * mtr::kDTX * 4 + (size_t)a.seg_cap * (KT * 4 + 16) +
                     (size_t)(KT + 3) * mtr::kDTX * 4;
  auto kern = mtr::detector_stream_kernel<KT>;
  if (lds > 160 * 1024) return kDetNoStream;
  if (!stream_kernel_lds_starts_at_zero(kern)) return kDetNoStream;
  const int rc = mtr::allow_dynamic_lds((const void*)kern, lds);
  if (rc != MTR_OK) return rc;
  long long per_strip = (long long)a.planes * a.n_seg;
  const int grid = (int)((per_strip < G ? per_strip : G) * tiles_x);
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, stream, a, detector_lut());
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

template <int KT>
static int launch_detector_pre(const uint8_t* images_u8, int N, int H, int W, const mtr_detector_geom* g,
                               int ty, int rows_cap, int pitch, float* out, hipStream_t stream) {
  if (((size_t)N * 3 * H * W) % 16)
    return launch_detector_pre_t<KT, true>(images_u8, N, H, W, g, ty, rows_cap, pitch, out, stream);
  return launch_detector_pre_t<KT, false>(images_u8, N, H, W, g, ty, rows_cap, pitch, out, stream);
}

extern "C" int mtr_detector_preprocess(const uint8_t* images_u8, int N, int H, int W,
                                       const mtr_detector_geom* g, float* out, mtr_stream_t stream) {
  return mtr_detector_preprocess_kernel(images_u8, N, H, W, g, MTR_DETECTOR_KERNEL_AUTO, out, stream);
}

extern "C" int mtr_detector_preprocess_kernel(const uint8_t* images_u8, int N, int H, int W,
                                              const mtr_detector_geom* g, int kernel, float* out,
                                              mtr_stream_t stream) {
  if (!images_u8 || !g || !out) return MTR_E_NULL;
  if (kernel < MTR_DETECTOR_KERNEL_AUTO || kernel > MTR_DETECTOR_KERNEL_STREAM) return MTR_E_PARAM;
  if (N < 0 || H <= 0 || W <= 0) return MTR_E_SHAPE;
  if (g->target_h <= 0 || g->target_w <= 0 || g->out_h < g->target_h || g->out_w < g->target_w ||
      g->pad_top < 0 || g->pad_left < 0 || g->pad_top + g->target_h > g->out_h ||
      g->pad_left + g->target_w > g->out_w)
    return MTR_E_PARAM;
  if ((uintptr_t)images_u8 % 16) return MTR_E_ALIGN;
  if (N == 0) return MTR_OK;
  const float sx = (float)W / (float)g->target_w, sy = (float)H / (float)g->target_h;
  const float supx = (g->antialias && sx >= 1.0f) ? sx : 1.0f, supy = (g->antialias && sy >= 1.0f) ? sy : 1.0f;
  const int taps = (int)(2.0f * (supx > supy ? supx : supy)) + 2;  // isize <= 2 * support + 1
  if (taps > mtr::kDTaps) return MTR_E_SHAPE;  // > 19x shrink: resize in two steps
  // bytes of a staged row: 64 columns' worth of source + both supports + alignment slack
  const int pitch = (((int)(mtr::kDTX * (sx > 1.0f ? sx : 1.0f) + 2.0f * supx) + 40) + 15) / 16 * 16;
  // rows a tile of TY output rows needs: TY * scale + 2 * support + 2; fit stage + temp in 60 KiB
  // (budget 60 KiB keeps 2-3 workgroups per CU; extreme shrinks on both axes take up to 120 KiB
  //  for a single output row per tile rather than being rejected)
  int ty = mtr::kDTYMax, rows_cap = 0;
  for (size_t budget : {(size_t)60 * 1024, (size_t)120 * 1024}) {
    for (ty = mtr::kDTYMax; ty >= 1; --ty) {
      rows_cap = (int)((float)ty * (sy > 1.0f ? sy : 1.0f) + 2.0f * supy) + 3;
      if ((size_t)rows_cap * (pitch + mtr::kDTX * sizeof(float)) <= budget) break;
    }
    if (ty >= 1) break;
  }
  if (ty < 1) return MTR_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int kt = !g->antialias ? 2 : taps <= 12 ? 12 : taps <= 24 ? 24 : mtr::kDTaps;
  if (kernel == MTR_DETECTOR_KERNEL_AUTO) {
    // The streaming kernel has ~5 us more set-up (64 KiB table, the loading wave's first chunks) and
    // half the cost per byte: measured cross-over (tools/experiments/detector_sizes.py) at 4 frames
    // of 1080p, 5 of 480 x 640; always ahead from 5.5x shrinks on (one 2160p frame: 39 vs 85 us),
    // behind on frames that are not shrunk (2 taps: nothing to win).
    const double in_mb = (double)N * 3 * H * W * 1e-6, out_m = (double)N * 3 * g->out_h * g->out_w * 1e-6;
    const bool stream = kt >= 24 || (kt == 12 && 0.21 * in_mb + 3.75 * out_m >= 8.5);
    kernel = stream ? MTR_DETECTOR_KERNEL_AUTO : MTR_DETECTOR_KERNEL_TILE;
  }
  if (kernel != MTR_DETECTOR_KERNEL_TILE) {
    int rc = MTR_E_SHAPE;
    if (kt == 2) rc = launch_detector_stream<2>(images_u8, N, H, W, g, pitch, out, s);
    else if (kt == 12) rc = launch_detector_stream<12>(images_u8, N, H, W, g, pitch, out, s);
    else if (kt == 24) rc = launch_detector_stream<24>(images_u8, N, H, W, g, pitch, out, s);
    else rc = launch_detector_stream<mtr::kDTaps>(images_u8, N, H, W, g, pitch, out, s);
    if (rc != kDetNoStream) return rc;
    if (kernel == MTR_DETECTOR_KERNEL_STREAM) return MTR_E_SHAPE;  // asked for by name and not available
  }
  if (!g->antialias) return launch_detector_pre<2>(images_u8, N, H, W, g, ty, rows_cap, pitch, out, s);
  if (taps <= 12) return launch_detector_pre<12>(images_u8, N, H, W, g, ty, rows_cap, pitch, out, s);
  if (taps <= 24) return launch_detector_pre<24>(images_u8, N, H, W, g, ty, rows_cap, pitch, out, s);
  return launch_detector_pre<mtr::kDTaps>(images_u8, N, H, W, g, ty, rows_cap, pitch, out, s);
}

extern "C" int mtr_detector_scale_boxes(const float* xyxy_conf, int n, const mtr_detector_geom* g,
                                        float* boxes_out, mtr_stream_t stream) {
  if (n < 0) return MTR_E_SHAPE;
  if (n == 0) return MTR_OK;
  if (!xyxy_conf || !g || !boxes_out) return MTR_E_NULL;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(mtr::detector_boxes_kernel, dim3((n + 127) / 128), dim3(128), 0,
                     (hipStream_t)stream, xyxy_conf, n, (float)g->pad_left, (float)g->pad_top,
                     g->x_factor, g->y_factor, boxes_out);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}
