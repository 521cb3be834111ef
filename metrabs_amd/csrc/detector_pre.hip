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
#include "common.h"
#include "resize_aa.h"

// developer-only timing ablations (tools/experiments/ablate_detector.py); 0 in the product
#ifndef MTR_DET_ABLATE
#define MTR_DET_ABLATE 0
#endif

namespace mtr {

constexpr int kDTX = 64;     // output columns per tile
constexpr int kDTYMax = 8;   // output rows per tile (host picks <= this)
// dynamic LDS: stage [rows_cap][pitch] uint8, then temp [rows_cap][64] f32
template <int KT, bool TAIL>
__global__ __launch_bounds__(256) void detector_pre_kernel(
    const uint8_t* __restrict__ src, size_t src_bytes, int planes, AxisGeom gx, AxisGeom gy,
    int pad_top, int pad_left, int out_h, int out_w, int ty_rows, int rows_cap, int pitch,
    float* __restrict__ out) {
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
    const float v = (float)pow((double)__fdiv_rn((float)tid, 255.0f), (double)2.2f);
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
        out[((size_t)pl * out_h + py) * out_w + px] = live[u] ? powf(acc[u], (float)(1.0 / 2.2)) : 0.5f;
      }
    }
  }
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
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cache[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cache[dev] = n;
  }
  return cache[dev];
}

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
                     gx, gy, g->pad_top, g->pad_left, g->out_h, g->out_w, ty, rows_cap, pitch, out);
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
  if (!images_u8 || !g || !out) return MTR_E_NULL;
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
