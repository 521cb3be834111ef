// K11: depthwise 3x3 convolution with the K10 epilogue (bias + activation, optionally the per-plane
// mean for the squeeze-excite block behind it) in one pass, NCHW.
//
// Like K10 this sits outside the reference's hot path: it serves the
// PyTorch-ROCm backbone's inference copy (backbones.fold_batchnorm(fused_epilogue=True)).  MIOpen
// runs these layers with its naive direct kernel and PyTorch's own depthwise kernel takes ~61 us
// per layer at the bench shape, followed by the bias / activation / mean passes; here a (b, c)
// plane is read once and written once: HBM-bound, the taps re-read the plane out of L1.
#include "common.h"

namespace mtr {

// Work item of a wave = 64 groups of four horizontally adjacent outputs: one 16x16 plane, four
// 8x8 planes, a quarter of a 32x32 plane.  A (b, c) plane is 0.25 - 4 KB, so a wave that handles
// one item and exits spends its life in two dependent latencies (weights, then taps): the first
// version ran at 38 % of the HBM spec.  Here the grid is persistent (a few waves per SIMD) and
// every wave walks its items with the raw taps AND weights of the next two items already in
// flight (a ring of three register sets, the loop unrolled over it); zeroing of out-of-plane taps
// happens when an item is finished, not when it is loaded, so that nothing waits on a load that
// was just issued.  Rows with pad 1 and W % 4 == 0 (every layer of the backbones at the shipped
// resolutions) are read as aligned vectors: stride 1 = one vector + two edge elements per row,
// stride 2 = two vectors + the left edge element.
struct DwGeom {
  int n_planes;
  int C, H, W, OH, OW;
  int pad, pad_left;  // top / left zero padding (bottom / right follow from OH, OW)
  int lpp;        // lanes per plane inside an item (power of two <= 64)
  int chunks;     // items per plane group (> 1: planes of more than 64 groups)
  int groups;     // groups of four outputs per plane = OH * OW / 4
  int n_pg;       // plane groups = ceil(n_planes / (64 / lpp))
  float inv_hw;
};

template <typename T, int STRIDE>
struct DwRaw {
  static constexpr int NV = STRIDE;  // aligned 4-element vectors per row
  struct alignas(4 * sizeof(T)) V4 { T v[4]; };
  V4 mid[3][NV];
  T edge[3][2];          // [left, right]; stride 2 uses the left one only
  T tap[3][3 * STRIDE + 3];  // scalar path (pad 0 or unaligned rows)
  float w[9], b;
};

template <typename T, int STRIDE, bool VEC>
__device__ __forceinline__ void dw_issue(const T* __restrict__ x, const float* __restrict__ w,
                                         const float* __restrict__ bias, const DwGeom& g, int p,
                                         int v, DwRaw<T, STRIDE>& r) {
  using Raw = DwRaw<T, STRIDE>;
  constexpr int NIN = 3 * STRIDE + 3;
  const int c = (int)((unsigned)p % (unsigned)g.C);
#pragma unroll
  for (int k = 0; k < 9; ++k) r.w[k] = w[c * 9 + k];
  r.b = bias[c];
  const int ow4 = g.OW >> 2;
  const int oy = v / ow4, ox0 = (v - oy * ow4) << 2;
  const T* xp = x + (size_t)p * (size_t)(g.H * g.W);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    int iy = oy * STRIDE - g.pad + ky;
    iy = iy < 0 ? 0 : (iy >= g.H ? g.H - 1 : iy);  // clamped: masked when the item is finished
    const T* row = xp + iy * g.W;
    if constexpr (VEC) {
      // the 4 STRIDE aligned columns from ix0 on + one edge element: column ix0 - 1 with left
      // padding 1; with left padding 0 (stride 2 behind TF-'SAME' / bottom-right padding) column
      // ix0 + 8 on the right
      const int ix0 = ox0 * STRIDE;
#pragma unroll
      for (int n = 0; n < Raw::NV; ++n)
        r.mid[ky][n] = *reinterpret_cast<const typename Raw::V4*>(row + ix0 + 4 * n);
      if constexpr (STRIDE == 1) {
        r.edge[ky][0] = row[ix0 > 0 ? ix0 - 1 : 0];
        r.edge[ky][1] = row[ix0 + 4 < g.W ? ix0 + 4 : 0];
      } else {
        const int ie = g.pad_left ? ix0 - 1 : ix0 + 8;
        r.edge[ky][0] = row[ie < 0 ? 0 : (ie >= g.W ? g.W - 1 : ie)];
      }
    } else {
#pragma unroll
      for (int j = 0; j < NIN; ++j) {
        int ix = ox0 * STRIDE - g.pad_left + j;
        ix = ix < 0 ? 0 : (ix >= g.W ? g.W - 1 : ix);
        r.tap[ky][j] = row[ix];
      }
    }
  }
}

template <typename T, int ACT, int STRIDE, bool VEC>
__device__ __forceinline__ float dw_finish(T* __restrict__ y, const DwGeom& g, int p, int v,
                                           bool live, const DwRaw<T, STRIDE>& r) {
  constexpr int NIN = 3 * STRIDE + 3;
  struct alignas(4 * sizeof(T)) Out4 { T v[4]; };
  const int ow4 = g.OW >> 2;
  const int oy = v / ow4, ox0 = (v - oy * ow4) << 2;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * STRIDE - g.pad + ky;
    const bool row_ok = iy >= 0 && iy < g.H;
    float in[NIN];
    if constexpr (VEC) {
      const int ix0 = ox0 * STRIDE;
      // in[j] = column ix0 - pad_left + j
      const int m0 = (STRIDE == 1 || g.pad_left) ? 1 : 0;  // where the aligned vectors start
#pragma unroll
      for (int n = 0; n < STRIDE; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = row_ok ? to_f32(r.mid[ky][n].v[j]) : 0.0f;
          if (STRIDE == 1 || g.pad_left) in[1 + 4 * n + j] = v; else in[4 * n + j] = v;
        }
      if constexpr (STRIDE == 1) {
        in[0] = (row_ok && ix0 > 0) ? to_f32(r.edge[ky][0]) : 0.0f;
        in[5] = (row_ok && ix0 + 4 < g.W) ? to_f32(r.edge[ky][1]) : 0.0f;
      } else if (m0) {
        in[0] = (row_ok && ix0 > 0) ? to_f32(r.edge[ky][0]) : 0.0f;
      } else {
        in[8] = (row_ok && ix0 + 8 < g.W) ? to_f32(r.edge[ky][0]) : 0.0f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NIN; ++j) {
        const int ix = ox0 * STRIDE - g.pad_left + j;
        in[j] = (row_ok && ix >= 0 && ix < g.W) ? to_f32(r.tap[ky][j]) : 0.0f;
      }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[o] = fmaf(in[o * STRIDE + kx], r.w[ky * 3 + kx], acc[o]);
  }
  Out4 out;
  float sum = 0.0f;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const float a = activate<ACT>(acc[o] + r.b);
    if constexpr (sizeof(T) == 4) out.v[o] = a; else out.v[o] = T(a);
    sum += to_f32(out.v[o]);
  }
  if (live) *reinterpret_cast<Out4*>(y + (size_t)p * (size_t)(g.OH * g.OW) + oy * g.OW + ox0) = out;
  return live ? sum : 0.0f;
}

template <typename T, int ACT, int STRIDE, bool VEC>
__global__ __launch_bounds__(256) void depthwise3x3_kernel(
    const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    T* __restrict__ y, float* __restrict__ row_mean, DwGeom g) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & (g.lpp - 1), pl = lane / g.lpp, ppi = 64 / g.lpp;
  const int n_waves = (int)gridDim.x * (int)(blockDim.x >> 6);
  const int wave = (int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
  // items of this wave, in order: plane groups wave, wave + n_waves, ...; inside a group its chunks
  int pg_i[3];
  int ch_i[3];
  auto item_plane = [&](int pg, int& p, bool& live) {
    const int plane = pg * ppi + pl;
    live = pg < g.n_pg && plane < g.n_planes;
    p = live ? plane : g.n_planes - 1;  // dead lanes / items recompute the last plane, no store
  };
  auto item_group = [&](int ch, bool& live) {
    const int v = ch * 64 + sub;  // (lpp < 64: one chunk, v = sub)
    if (v >= g.groups) live = false;
    return v < g.groups ? v : g.groups - 1;
  };
  auto advance = [&](int& pg, int& ch) {
    if (++ch == g.chunks) { ch = 0; pg += n_waves; }
  };
  DwRaw<T, STRIDE> r0, r1, r2;
  int pg = wave;
  int ch = 0;
  auto issue = [&](DwRaw<T, STRIDE>& r, int slot) {
    pg_i[slot] = pg; ch_i[slot] = ch;
    int p; bool live;
    item_plane(pg, p, live);
    const int v = item_group(ch, live);
    dw_issue<T, STRIDE, VEC>(x, w, bias, g, p, v, r);
    advance(pg, ch);
  };
  float sum = 0.0f;
  auto finish = [&](const DwRaw<T, STRIDE>& r, int slot) {
    int p; bool live;
    item_plane(pg_i[slot], p, live);
    const int v = item_group(ch_i[slot], live);
    sum += dw_finish<T, ACT, STRIDE, VEC>(y, g, p, v, live, r);
    if (row_mean && ch_i[slot] == g.chunks - 1) {  // (wave-uniform) the plane group is complete
      float s = sum;
      for (int m = g.lpp >> 1; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
      bool alive; int pp;
      item_plane(pg_i[slot], pp, alive);
      if (alive && sub == 0) row_mean[pp] = s * g.inv_hw;
      sum = 0.0f;
    }
  };
  if (wave >= g.n_pg) return;  // (whole wave)
  issue(r0, 0);
  issue(r1, 1);
  // items of this wave: ceil((n_pg - wave) / n_waves) * chunks
  const int n_items = ((g.n_pg - wave + n_waves - 1) / n_waves) * g.chunks;
  for (int i = 0; i < n_items; i += 3) {
    issue(r2, 2);
    finish(r0, 0);
    if (i + 1 >= n_items) break;
    issue(r0, 0);
    finish(r1, 1);
    if (i + 2 >= n_items) break;
    issue(r1, 1);
    finish(r2, 2);
  }
}

// ---- stride 1, pad 1, whole 4x4 output blocks (the 27 stride-1 layers of EfficientNetV2-S at 256 px:
// 16x16 and 8x8 planes).  The generic kernel above spends ~800 VALU instructions per lane and item
// (index arithmetic, 18 selects, per-group set-up) and is VALU-bound at 30 - 38 % of the HBM spec.
// Here a lane owns a 4 x 4 block of outputs: six aligned row vectors are all it loads; the columns
// left and right of them come from the neighbouring lanes through DPP row shifts (a plane row's
// lanes are adjacent and never straddle a 16-lane DPP row); rows above / below the plane and the
// plane's left / right edge are zeroed with 12 + 12 selects for 16 outputs; the lanes of a plane
// (1 .. 64, a power of two) reduce the squeeze-excite mean with xor shuffles.
template <int CTRL>
__device__ __forceinline__ float dpp_row_shift(float v) {
  // row_shr:1 (0x111): lane i gets lane i - 1; row_shl:1 (0x101): lane i gets lane i + 1;
  // bound_ctrl: lanes without a source inside the 16-lane row get 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void depthwise3x3_s1_block_kernel(
    const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    T* __restrict__ y, float* __restrict__ row_mean, int n_planes, int C, int H, int W, int lp_log2,
    int tw_log2, float inv_hw) {
  struct alignas(4 * sizeof(T)) V4 { T v[4]; };
  const int gid = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const int plane = gid >> lp_log2, local = gid & ((1 << lp_log2) - 1);
  const int ty = local >> tw_log2, tx = local & ((1 << tw_log2) - 1);
  const bool live = plane < n_planes;
  const int p = live ? plane : n_planes - 1;  // dead lanes recompute the last plane, no store
  const int c = (int)((unsigned)p % (unsigned)C);
  const int oy0 = ty << 2, ox0 = tx << 2;
  const T* xp = x + (size_t)p * (size_t)(H * W) + ox0;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  V4 raw[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    int iy = oy0 - 1 + r;
    iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
    if constexpr (sizeof(T) == 4) {  // (a native vector type: one global_load_dwordx4)
      const f32x4 t = *reinterpret_cast<const f32x4*>(xp + iy * W);
      raw[r].v[0] = t[0]; raw[r].v[1] = t[1]; raw[r].v[2] = t[2]; raw[r].v[3] = t[3];
    } else {
      raw[r] = *reinterpret_cast<const V4*>(xp + iy * W);
    }
  }
  float wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = w[c * 9 + k];
  const float b = bias[c];
  const bool has_left = tx > 0, has_right = tx + 1 < (1 << tw_log2);
  float in[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const bool row_ok = (r != 0 || oy0 > 0) && (r != 5 || oy0 + 4 < H);
#pragma unroll
    for (int j = 0; j < 4; ++j) in[r][1 + j] = row_ok ? to_f32(raw[r].v[j]) : 0.0f;
    const float l = dpp_row_shift<0x111>(in[r][4]), rgt = dpp_row_shift<0x101>(in[r][1]);
    in[r][0] = has_left ? l : 0.0f;
    in[r][5] = has_right ? rgt : 0.0f;
  }
  float sum = 0.0f;
  T* yp = y + (size_t)p * (size_t)(H * W) + oy0 * W + ox0;
#pragma unroll
  for (int orow = 0; orow < 4; ++orow) {
    float acc[4] = {b, b, b, b};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] = fmaf(in[orow + ky][o + kx], wk[ky * 3 + kx], acc[o]);
    V4 out;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const float a = activate<ACT>(acc[o]);
      if constexpr (sizeof(T) == 4) out.v[o] = a; else out.v[o] = T(a);
      sum += to_f32(out.v[o]);
    }
    if (live) {
      if constexpr (sizeof(T) == 4)
        *reinterpret_cast<f32x4*>(yp + orow * W) = f32x4{out.v[0], out.v[1], out.v[2], out.v[3]};
      else
        *reinterpret_cast<V4*>(yp + orow * W) = out;
    }
  }
  if (row_mean) {
    for (int m = (1 << lp_log2) >> 1; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
    if (live && local == 0) row_mean[plane] = sum * inv_hw;
  }
}

static int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

template <typename T, int STRIDE>
static int launch_depthwise(const void* x, const float* w, const float* bias, int act, void* y,
                            float* row_mean, long long n_planes, int C, int H, int W, int OH, int OW,
                            int pad, int pad_left, bool symmetric, hipStream_t stream) {
  if (n_planes >= (1LL << 30)) return MTR_E_SHAPE;
  if constexpr (STRIDE == 1) {
    const int tw = ilog2_exact(W / 4), th = ilog2_exact(H / 4);
    if (symmetric && pad == 1 && (W & 3) == 0 && (H & 3) == 0 && tw >= 0 && tw <= 4 && th >= 0 && tw + th <= 6 &&
        ((uintptr_t)x % 16) == 0 && n_planes < (1LL << 24)) {
      const int lp_log2 = tw + th;
      const long long lanes = n_planes << lp_log2;
      const dim3 grid((unsigned)((lanes + 255) / 256)), block(256);
      const float inv = 1.0f / (float)(H * W);
      MTR_CLEAR_STALE();
#define MTR_DW_BLOCK(A)                                                                            \
  hipLaunchKernelGGL((depthwise3x3_s1_block_kernel<T, A>), grid, block, 0, stream, (const T*)x, w, \
                     bias, (T*)y, row_mean, (int)n_planes, C, H, W, lp_log2, tw, inv)
      switch (act) {
        case kActNone: MTR_DW_BLOCK(kActNone); break;
        case kActRelu: MTR_DW_BLOCK(kActRelu); break;
        case kActSilu: MTR_DW_BLOCK(kActSilu); break;
        case kActHardswish: MTR_DW_BLOCK(kActHardswish); break;
        default: return MTR_E_PARAM;
      }
#undef MTR_DW_BLOCK
      MTR_CHECK_LAUNCH();
      return MTR_OK;
    }
  }
  DwGeom g;
  g.n_planes = (int)n_planes; g.C = C; g.H = H; g.W = W; g.OH = OH; g.OW = OW; g.pad = pad;
  g.pad_left = pad_left;
  g.groups = OH * (OW / 4);
  g.lpp = 1;
  while (g.lpp < 64 && g.lpp < g.groups) g.lpp <<= 1;
  g.chunks = (g.groups + 63) / 64;
  g.n_pg = (int)((n_planes + 64 / g.lpp - 1) / (64 / g.lpp));
  g.inv_hw = 1.0f / (float)(OH * OW);
  // persistent grid: 256 CUs x 4 SIMDs x 4 waves (three register sets of taps + weights per wave)
  int blocks = (g.n_pg + 3) / 4;
  if (blocks > 256 * 4) blocks = 256 * 4;
  const dim3 grid((unsigned)blocks), block(256);
  // aligned row vectors: stride 1 with symmetric padding 1; stride 2 when the 8 columns from
  // 2 ox0 on exist for every group (OW * 2 == W), left padding 1 or 0
  const bool vec = (W & 3) == 0 && ((uintptr_t)x % 16) == 0 &&
                   (STRIDE == 1 ? (symmetric && pad == 1) : (OW * 2 == W && pad_left <= 1));
  MTR_CLEAR_STALE();
#define MTR_DW_LAUNCH(A)                                                                            \
  if (vec)                                                                                          \
    hipLaunchKernelGGL((depthwise3x3_kernel<T, A, STRIDE, true>), grid, block, 0, stream, (const T*)x, \
                       w, bias, (T*)y, row_mean, g);                                                \
  else                                                                                              \
    hipLaunchKernelGGL((depthwise3x3_kernel<T, A, STRIDE, false>), grid, block, 0, stream,          \
                       (const T*)x, w, bias, (T*)y, row_mean, g)
  switch (act) {
    case kActNone: MTR_DW_LAUNCH(kActNone); break;
    case kActRelu: MTR_DW_LAUNCH(kActRelu); break;
    case kActSilu: MTR_DW_LAUNCH(kActSilu); break;
    case kActHardswish: MTR_DW_LAUNCH(kActHardswish); break;
    default: return MTR_E_PARAM;
  }
#undef MTR_DW_LAUNCH
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

template <typename T>
static int dispatch_depthwise(const void* x, const float* w, const float* bias, int act, void* y,
                              float* row_mean, long long n_planes, int C, int H, int W, int OH,
                              int OW, int stride, int pad, int pad_left, bool symmetric,
                              hipStream_t stream) {
  if (stride == 1)
    return launch_depthwise<T, 1>(x, w, bias, act, y, row_mean, n_planes, C, H, W, OH, OW, pad, pad_left,
                                  symmetric, stream);
  return launch_depthwise<T, 2>(x, w, bias, act, y, row_mean, n_planes, C, H, W, OH, OW, pad, pad_left,
                                symmetric, stream);
}

}  // namespace mtr

extern "C" int mtr_depthwise3x3_bias_act_padded(const void* x, int dtype, const float* weight,
                                                const float* bias, int act, long long B, int C, int H,
                                                int W, int stride, int pad_top, int pad_left,
                                                int pad_bottom, int pad_right, void* y, float* row_mean,
                                                mtr_stream_t stream) {
  if (!x || !weight || !bias || !y) return MTR_E_NULL;
  if (B < 0 || C <= 0 || H <= 0 || W <= 0) return MTR_E_SHAPE;
  if (stride != 1 && stride != 2) return MTR_E_PARAM;
  if (pad_top < 0 || pad_top > 1 || pad_left < 0 || pad_left > 1 || pad_bottom < 0 || pad_bottom > 2 ||
      pad_right < 0 || pad_right > 2)
    return MTR_E_PARAM;
  if (H + pad_top + pad_bottom < 3 || W + pad_left + pad_right < 3) return MTR_E_SHAPE;
  const int OH = (H + pad_top + pad_bottom - 3) / stride + 1, OW = (W + pad_left + pad_right - 3) / stride + 1;
  if (OW % 4 != 0) return MTR_E_SHAPE;
  if ((uintptr_t)y % 16) return MTR_E_ALIGN;
  if (B == 0) return MTR_OK;
  const bool symmetric = pad_top == pad_left && pad_bottom == pad_top && pad_right == pad_top;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case MTR_F32: return mtr::dispatch_depthwise<float>(x, weight, bias, act, y, row_mean, B * C, C, H, W, OH, OW, stride, pad_top, pad_left, symmetric, s);
    case MTR_F16: return mtr::dispatch_depthwise<__half>(x, weight, bias, act, y, row_mean, B * C, C, H, W, OH, OW, stride, pad_top, pad_left, symmetric, s);
    case MTR_BF16: return mtr::dispatch_depthwise<__hip_bfloat16>(x, weight, bias, act, y, row_mean, B * C, C, H, W, OH, OW, stride, pad_top, pad_left, symmetric, s);
    default: return MTR_E_DTYPE;
  }
}

extern "C" int mtr_depthwise3x3_bias_act(const void* x, int dtype, const float* weight,
                                         const float* bias, int act, long long B, int C, int H, int W,
                                         int stride, int pad, void* y, float* row_mean,
                                         mtr_stream_t stream) {
  if (pad != 0 && pad != 1) return (!x || !weight || !bias || !y) ? MTR_E_NULL : MTR_E_PARAM;
  return mtr_depthwise3x3_bias_act_padded(x, dtype, weight, bias, act, B, C, H, W, stride, pad, pad, pad,
                                          pad, y, row_mean, stream);
}
