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
// One workgroup walks output tiles of TY x 64 (persistent grid): per tile it derives the weights of
// its 64 columns and TY rows, runs the horizontal pass for the input rows the tile needs straight
// from the uint8 frame (gamma LUT in LDS) into an LDS intermediate, then the vertical pass, the
// re-gamma and the store; pad pixels are written as 0.5 by the same tile walk.
// Bound: HBM read of the uint8 frames (6.2 MB per 1080p frame) -- in practice LDS/ALU-limited, the
// taps of neighbouring outputs overlap 2x in each direction.
#include "common.h"

namespace mtr {

constexpr int kDTX = 64;     // output columns per tile
constexpr int kDTYMax = 8;   // output rows per tile (host picks <= this)
constexpr int kDTaps = 40;   // taps per output index: ceil(2 * scale) + 2 <= 40 (scale <= 19)
constexpr int kDRows = 128;  // input rows of the LDS intermediate per tile

struct AxisGeom {
  int in_size, out_size;  // frame / resized extent along this axis
  int aa;                 // antialias (shrinking) or plain bilinear
};

// aten/native/cpu/UpSampleKernel.cpp: _compute_indices_min_size_weights_aa (antialias) and
// compute_indices_weights / guard_index_and_lambda (plain).  `w` has stride `ws` floats.
__device__ __forceinline__ void axis_weights(int i, const AxisGeom& g, int& imin, int& isize, float* w,
                                             int ws) {
  const float scale = (float)g.in_size / (float)g.out_size;
  if (g.aa) {
    const float support = scale >= 1.0f ? scale : 1.0f;
    const float invscale = scale >= 1.0f ? (float)(1.0 / (double)scale) : 1.0f;
    const float center = (float)((double)scale * ((double)i + 0.5));
    long long lo = (long long)((double)(center - support) + 0.5);
    long long hi = (long long)((double)(center + support) + 0.5);
    if (lo < 0) lo = 0;
    if (hi > g.in_size) hi = g.in_size;
    imin = (int)lo;
    isize = (int)(hi - lo);
    if (isize > kDTaps) isize = kDTaps;  // (host rejects scales that need more)
    float total = 0.0f;
    for (int j = 0; j < isize; ++j) {
      float x = (float)(((double)((float)(j + imin) - center) + 0.5) * (double)invscale);
      x = x < 0.0f ? -x : x;
      const float wj = x < 1.0f ? 1.0f - x : 0.0f;
      w[j * ws] = wj;
      total = __fadd_rn(total, wj);
    }
    if (total != 0.0f)
      for (int j = 0; j < isize; ++j) w[j * ws] = __fdiv_rn(w[j * ws], total);
  } else {
    float real = (float)((double)scale * ((double)i + 0.5) - 0.5);
    if (real < 0.0f) real = 0.0f;
    int i0 = (int)floorf(real);
    if (i0 > g.in_size - 1) i0 = g.in_size - 1;
    float l1 = real - (float)i0;
    l1 = fminf(fmaxf(l1, 0.0f), 1.0f);
    imin = i0;
    isize = 2;  // tap 1 is read at min(i0 + 1, in_size - 1)
    w[0] = 1.0f - l1;
    w[ws] = l1;
  }
}

__global__ __launch_bounds__(256) void detector_pre_kernel(
    const uint8_t* __restrict__ src, int planes, AxisGeom gx, AxisGeom gy, int pad_top, int pad_left,
    int out_h, int out_w, int ty_rows, float* __restrict__ out) {
  __shared__ float lut[256];
  __shared__ float wx[kDTaps][kDTX];        // [tap][column]: conflict-free across columns
  __shared__ float wy[kDTYMax][kDTaps];
  __shared__ int xmin[kDTX], xsize[kDTX], ymin[kDTYMax], ysize[kDTYMax];
  __shared__ float temp[kDRows][kDTX];
  __shared__ int yrange[2];

  const int tid = threadIdx.x;
  lut[tid] = (float)pow((double)__fdiv_rn((float)tid, 255.0f), (double)2.2f);

  const int tiles_x = (out_w + kDTX - 1) / kDTX, tiles_y = (out_h + ty_rows - 1) / ty_rows;
  const long long n_tiles = (long long)planes * tiles_y * tiles_x;
  const int c = tid & (kDTX - 1), rg = tid >> 6;  // column of the tile, row group 0..3

  for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int tx = (int)(t % tiles_x), tyi = (int)((t / tiles_x) % tiles_y);
    const int pl = (int)(t / ((long long)tiles_x * tiles_y));
    __syncthreads();  // previous tile fully consumed (and the LUT written, first time)
    // ---- weights of this tile's columns (threads 0..63) and rows (threads 64..64+TY-1)
    if (tid < kDTX) {
      const int ox = tx * kDTX + tid - pad_left;  // column in the resized image
      int mn = 0, sz = 0;
      if (ox >= 0 && ox < gx.out_size) axis_weights(ox, gx, mn, sz, &wx[0][tid], kDTX);
      xmin[tid] = mn;
      xsize[tid] = sz;  // 0 = pad column
    } else if (tid < kDTX + ty_rows) {
      const int r = tid - kDTX;
      const int oy = tyi * ty_rows + r - pad_top;
      int mn = 0, sz = 0;
      if (oy >= 0 && oy < gy.out_size) axis_weights(oy, gy, mn, sz, &wy[r][0], 1);
      ymin[r] = mn;
      ysize[r] = sz;
    }
    __syncthreads();
    if (tid == 0) {
      int lo = 0x7fffffff, hi = 0;
      for (int r = 0; r < ty_rows; ++r)
        if (ysize[r] > 0) {
          lo = min(lo, ymin[r]);
          hi = max(hi, min(ymin[r] + ysize[r], gy.in_size));
        }
      yrange[0] = lo;
      yrange[1] = hi > lo ? hi - lo : 0;
    }
    __syncthreads();
    const int y_lo = yrange[0], n_rows = min(yrange[1], kDRows);
    // ---- horizontal pass: input rows y_lo .. y_lo + n_rows - 1, this thread's column
    const int xs = xsize[c], xm = xmin[c];
    if (xs > 0) {
      const uint8_t* plane = src + (size_t)pl * gy.in_size * gx.in_size;
      for (int r = rg; r < n_rows; r += 4) {
        const uint8_t* row = plane + (size_t)(y_lo + r) * gx.in_size;
        float acc;
        if (gx.aa) {
          acc = __fmul_rn(lut[row[xm]], wx[0][c]);
          for (int j = 1; j < xs; ++j) acc = __fmaf_rn(lut[row[xm + j]], wx[j][c], acc);
        } else {
          const float v0 = lut[row[xm]], v1 = lut[row[min(xm + 1, gx.in_size - 1)]];
          acc = __fmaf_rn(v0, wx[0][c], __fmul_rn(v1, wx[1][c]));
        }
        temp[r][c] = acc;
      }
    }
    __syncthreads();
    // ---- vertical pass + re-gamma + store (pad pixels: 0.5)
    for (int r = rg; r < ty_rows; r += 4) {
      const int py = tyi * ty_rows + r, px = tx * kDTX + c;  // position in the padded output
      if (py >= out_h || px >= out_w) continue;
      float v = 0.5f;
      const int ys = ysize[r];
      if (ys > 0 && xs > 0) {
        const int y0 = ymin[r] - y_lo;
        float acc;
        if (gy.aa) {
          acc = __fmul_rn(temp[y0][c], wy[r][0]);
          for (int k = 1; k < ys; ++k) acc = __fmaf_rn(temp[y0 + k][c], wy[r][k], acc);
        } else {
          const int y1 = min(ymin[r] + 1, gy.in_size - 1) - y_lo;
          acc = __fmaf_rn(temp[y0][c], wy[r][0], __fmul_rn(temp[y1][c], wy[r][1]));
        }
        v = powf(acc, (float)(1.0 / 2.2));
      }
      out[((size_t)pl * out_h + py) * out_w + px] = v;
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

extern "C" int mtr_detector_preprocess(const uint8_t* images_u8, int N, int H, int W,
                                       const mtr_detector_geom* g, float* out, mtr_stream_t stream) {
  if (!images_u8 || !g || !out) return MTR_E_NULL;
  if (N < 0 || H <= 0 || W <= 0) return MTR_E_SHAPE;
  if (g->target_h <= 0 || g->target_w <= 0 || g->out_h < g->target_h || g->out_w < g->target_w ||
      g->pad_top < 0 || g->pad_left < 0 || g->pad_top + g->target_h > g->out_h ||
      g->pad_left + g->target_w > g->out_w)
    return MTR_E_PARAM;
  if (N == 0) return MTR_OK;
  const float sx = (float)W / (float)g->target_w, sy = (float)H / (float)g->target_h;
  const float supx = (g->antialias && sx >= 1.0f) ? sx : 1.0f, supy = (g->antialias && sy >= 1.0f) ? sy : 1.0f;
  if (2.0f * supx + 2.0f > (float)mtr::kDTaps || 2.0f * supy + 2.0f > (float)mtr::kDTaps)
    return MTR_E_SHAPE;  // > 19x shrink: resize in two steps
  // rows of the intermediate a tile needs: TY * scale + 2 * support + 2 <= kDRows
  int ty = (int)(((float)mtr::kDRows - 2.0f * supy - 3.0f) / (sy > 1.0f ? sy : 1.0f));
  if (ty > mtr::kDTYMax) ty = mtr::kDTYMax;
  if (ty < 1) return MTR_E_SHAPE;
  const mtr::AxisGeom gx{W, g->target_w, g->antialias}, gy{H, g->target_h, g->antialias};
  const long long tiles = (long long)N * 3 * ((g->out_h + ty - 1) / ty) * ((g->out_w + mtr::kDTX - 1) / mtr::kDTX);
  const int grid = (int)(tiles < 256 * 3 ? tiles : 256 * 3);  // persistent, 3 workgroups per CU
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(mtr::detector_pre_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, images_u8,
                     N * 3, gx, gy, g->pad_top, g->pad_left, g->out_h, g->out_w, ty, out);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
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
