// K6: crop sampler -- gamma decode + box pyramid, per-box crop geometry, perspective/lens warp.
//
// Replaces (all in metrabs_pytorch/multiperson/):
//   multiperson_model.py:196        images = (u8/255)**2.2           -> build_pyramid_kernel (LUT)
//   warping.py:10-13                avg_pool2d 2x2 pyramid, 3 levels  -> build_pyramid_kernel
//   multiperson_model.py:322-355    _get_new_rotation_and_scale       -> crop_geometry_kernel
//   multiperson_model.py:264-305    new intrinsics, R, inv(K_new R)   -> crop_geometry_kernel
//   warping.py:15-21,128-133        per-level intrinsics, level pick   -> crop_geometry_kernel
//   warping.py:23-54                per-crop Python loop: meshgrid, 2 einsums, distort,
//                                   grid_sample(bilinear, zeros, align_corners=True)
//                                                                      -> warp_crops_kernel
//   multiperson_model.py:308-319    avg_pool2d(aa) and crops **= gamma/2.2 -> warp_crops_kernel
//
// The reference spends ~10 launches per crop in a Python loop (~100 crops/s on 8 CPU cores); here
// the whole internal batch is ONE launch.  Bound: HBM for the streaming output (3*res^2*sizeof(out)
// per crop) plus the clipped source footprint, which normally stays in L2 / Infinity Cache.
#include <type_traits>

#include "common.h"
#include "resize_aa.h"

// developer-only timing ablations of warp_crops_kernel (tools/experiments/ablate_warp.py); 0 in the product
#ifndef MTR_WARP_PX
#define MTR_WARP_PX 4  // output pixels per thread (a multiple of 4 for the vector stores)
#endif
#ifndef MTR_WARP_LX
#define MTR_WARP_LX 32       // lanes of a wave along x in warp_rows_kernel (64 / LX rows per iteration):
                             // 32 x 2 keeps a rotated crop's wave on fewer source lines than 64 x 1
                             // (320 TTA crops 157.6 -> 140.7 us, plain 64 crops 25.2 -> 23.9; 16 x 4: 145.9 / 25.6)
#endif
#ifndef MTR_WARP_PREFETCH
#define MTR_WARP_PREFETCH 1  // samples requested ahead of the one being finished
#endif
#ifndef MTR_WARP_RCP
#define MTR_WARP_RCP 1  // 1/oz by v_rcp_f32 + one Newton step (<= 1 ulp) instead of the IEEE division sequence
#endif
#ifndef MTR_WARP_LEAN
#define MTR_WARP_LEAN 1  // warp_rows_kernel, round 3: fewer VALU instructions per sample, same taps and the same
                         // weights bit for bit (checksums of the 64- and 320-crop outputs unchanged) -- tap-pair
                         // weights by clamps of d = ix - xs instead of selects on (x0 - xs) and the in-range test
                         // (an out-of-range or non-finite coordinate gives d outside (-1, 2): both weights 0); when
                         // the row and plane pitches are multiples of 4 the six loads of a sample share ONE address
                         // register, the plane and row offsets ride in the buffer instruction's scalar offset, and
                         // the byte window is one v_alignbyte.  64 crops, cache-resident frames: 25.6 -> 24.5 us.
#endif
#ifndef MTR_PYR_LUT_COPIES
#define MTR_PYR_LUT_COPIES 16   // copies of the gamma table in build_pyramid_u8_wide_kernel's LDS (= KiB per workgroup).  32 (one per
                                // bank: no conflict whatever the pixels) leaves five workgroups per CU; 16 (lanes l and l + 16 share a
                                // copy: two-way when their values differ and have the same parity) leaves eight, the wave limit --
                                // 8 x 1080p planar frames in rotation 22.7 -> 19.8 us (0.61 -> 0.70 of HBM), interleaved 21.3 -> 21.4;
                                // 8 copies the same as 16; same bits (profiles/r06zf_pyramid_lut.jsonl)
#endif
#ifndef MTR_WARP_WAVES
#define MTR_WARP_WAVES 4   // waves per workgroup of warp_rows_kernel (a workgroup = LX columns x WAVES * ROWS * 64 / LX rows)
#endif
#ifndef MTR_WARP_ASM
#define MTR_WARP_ASM 0   // warp_rows_kernel, round 6 (a developer option, measured and left off): the six tap loads of a sample issued
                         // from inline asm and awaited with a COUNTED s_waitcnt vmcnt tied to their registers.  The ISA of the
                         // builtin-load pipeline waits vmcnt(0) in front of every other sample -- for the loads of the NEXT
                         // sample and the previous pixel's stores too (the two alternative request paths meet in a join the
                         // compiler's wait counting gives up on) -- so "one sample requested ahead" drains every second
                         // sample.  With exact waits (vmcnt 6 / 9 / 9 / 3), same bits: 64 crops 28.2 -> 27.2 us planar, 25.0 ->
                         // 25.2 interleaved levels; 320 TTA crops 154.9 -> 158.4; two or three samples ahead: the same
                         // (profiles/r06u_warp_asm.jsonl).  Eight waves per SIMD already cover each other's waits: the
                         // kernel is not latency-bound on its taps.
#endif
#ifndef MTR_WARP_ABLATE
#define MTR_WARP_ABLATE 0   // developer-only timing ablations (tools/experiments/ablate_warp.py); in warp_rows_kernel:
                            // 1 = no tap loads, 2 = no LUT reads, 4 = no gamma pow, 8 = no stores, 16 = one gather pair for all channels.  Round 3, 64 crops,
                            // cache-resident / rotating frames: full 24.5 / 30.3 us, no taps 18.7 / 18.8, no stores
                            // 17.6 / 20.1, none of the four 11.1 / 11.0 -- arithmetic, taps and stores are nearly
                            // additive; neither fewer VALU instructions (this macro's LEAN) nor a quarter of the store
                            // instructions (16-byte stores through an LDS transpose: 24.7 us, not kept) moves the sum
#endif

namespace mtr {

// ------------------------------------------------------------------------------------------------
// pyramid: each thread owns a 4x4 block of level-0 pixels = 2x2 of level 1 = 1 pixel of level 2.
// avg_pool2d accumulates row-major ((a+b)+c)+d and divides by 4 (exact in fp32).
// WRITE_L0 = false: the f32 level 0 (64 % of the pyramid's bytes) is NOT materialised; the sampler
// reads level 0 straight from the uint8 frame through the 256-entry LUT exported to `lut_out`.
// HWC (uint8 only, level 0 not written): the frames are interleaved [N,H,W,3]; plane pl = image pl / 3,
// channel pl % 3, its pixels 3 bytes apart (the path of odd frame sizes: one byte per load).
template <bool FROM_U8, bool WRITE_L0, bool HWC = false>
__global__ __launch_bounds__(256) void build_pyramid_kernel(
    const void* __restrict__ src_any, int planes, int Hi, int Wi, float* __restrict__ l0,
    float* __restrict__ l1, float* __restrict__ l2, float* __restrict__ lut_out, GammaLut lut_in,
    FastDiv by_bw, FastDiv by_bh) {
  __shared__ float lut[256];
  if (FROM_U8) {
    lut[threadIdx.x] = lut_in.v[threadIdx.x];  // (v/255)**2.2, common.h
    if (lut_out != nullptr && blockIdx.x == 0) lut_out[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
  }
  const uint8_t* __restrict__ src = (const uint8_t*)src_any;
  const float* __restrict__ srcf = (const float*)src_any;

  const int H1 = Hi / 2, W1 = Wi / 2, H2 = H1 / 2, W2 = W1 / 2;
  const int bw = (Wi + 3) / 4, bh = (Hi + 3) / 4;
  const long long total = (long long)planes * bh * bw;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    int bx, by, pl;
    if (total <= 0xffffffffLL) {  // (wave-uniform) no 64-bit division per block: common.h FastDiv
      const unsigned rowi = fastdiv((unsigned)t, by_bw);
      bx = (int)((unsigned)t - rowi * (unsigned)bw);
      pl = (int)fastdiv(rowi, by_bh);
      by = (int)(rowi - (unsigned)pl * (unsigned)bh);
    } else {
      bx = (int)(t % bw);
      by = (int)((t / bw) % bh);
      pl = (int)(t / ((long long)bw * bh));
    }
    const int x0 = bx * 4, y0 = by * 4;
    const uint8_t* sp = HWC ? src + (size_t)(pl / 3) * 3 * Hi * Wi + (pl % 3) : src + (size_t)pl * Hi * Wi;
    constexpr int PS = HWC ? 3 : 1;
    const float* spf = srcf + (size_t)pl * Hi * Wi;
    float* d0 = l0 + (size_t)pl * Hi * Wi;
    float v[4][4];
    const bool fast = !HWC && (Wi % 4 == 0) && (x0 + 4 <= Wi) && (y0 + 4 <= Hi);
    if (!FROM_U8) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int y = y0 + r, x = x0 + c;
          v[r][c] = (y < Hi && x < Wi) ? spf[(size_t)y * Wi + x] : 0.0f;
        }
    } else if (fast) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t raw = *reinterpret_cast<const uint32_t*>(sp + (size_t)(y0 + r) * Wi + x0);
        v[r][0] = lut[raw & 0xff];
        v[r][1] = lut[(raw >> 8) & 0xff];
        v[r][2] = lut[(raw >> 16) & 0xff];
        v[r][3] = lut[raw >> 24];
        if (WRITE_L0)
          *reinterpret_cast<float4*>(d0 + (size_t)(y0 + r) * Wi + x0) =
              make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int y = y0 + r, x = x0 + c;
          float val = 0.0f;
          if (y < Hi && x < Wi) {
            val = lut[sp[((size_t)y * Wi + x) * PS]];
            if (WRITE_L0) d0[(size_t)y * Wi + x] = val;
          }
          v[r][c] = val;
        }
    }
    float q[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float s = __fadd_rn(__fadd_rn(__fadd_rn(v[2 * r][2 * c], v[2 * r][2 * c + 1]),
                                            v[2 * r + 1][2 * c]), v[2 * r + 1][2 * c + 1]);
        q[r][c] = s * 0.25f;
        const int y1 = by * 2 + r, x1 = bx * 2 + c;
        if (y1 < H1 && x1 < W1) l1[((size_t)pl * H1 + y1) * W1 + x1] = q[r][c];
      }
    if (by < H2 && bx < W2) {
      const float s = __fadd_rn(__fadd_rn(__fadd_rn(q[0][0], q[0][1]), q[1][0]), q[1][1]);
      l2[((size_t)pl * H2 + by) * W2 + bx] = s * 0.25f;
    }
  }
}

// Wide variant for uint8 frames whose width and height are multiples of 8 (every video format),
// level 0 not materialised: one thread owns an 8x8 tile = four of the blocks above, laid out so
// that EVERY access of a wave is lane-contiguous -- 8-byte loads of 8 pixels (512 B per wave and
// row), float4 stores to level 1 (1 KiB per wave and row), float2 stores to level 2.  (The 4x4
// kernel reads 4 B and writes 8 B / 4 B per lane; a 16x4 strip per thread had 16-byte loads but
// 32-byte-strided level-1 stores and was slower.)  Same values, same addition order.
// The gamma LUT is replicated MTR_PYR_LUT_COPIES times ([value][copies]; lane l reads copy l & (copies - 1)): with 32
// copies -- one per bank; lanes l and l+32 of a ds_read_b32 are serviced separately -- the lookups are conflict-free
// whatever the pixel values (random pixels averaged ~3.5 ways on the shared 256-entry table); the shipped 16 trade a
// two-way conflict on some lookups for eight workgroups per CU instead of five (see the macro).
// HWC: interleaved frames [N,H,W,3]; `planes` is then the number of IMAGES and a thread owns the 8x8
// tile of all three channels: 24 contiguous bytes per row (three 8-byte loads, lane-contiguous across
// the wave as 1,536 B per row), the same level-1 / level-2 stores into each channel's plane.
template <bool HWC>
__global__ __launch_bounds__(256) void build_pyramid_u8_wide_kernel(
    const uint8_t* __restrict__ src, int planes, int Hi, int Wi, float* __restrict__ l1,
    float* __restrict__ l2, float* __restrict__ lut_out, GammaLut lut_in, FastDiv by_tw, FastDiv by_th) {
  constexpr int COP = MTR_PYR_LUT_COPIES;  // copies of the table (32: one per bank)
  __shared__ __attribute__((aligned(16))) float lut[256 * COP];
  {
    const float v = lut_in.v[threadIdx.x];
    if (lut_out != nullptr && blockIdx.x == 0) lut_out[threadIdx.x] = v;
    const float4 v4 = make_float4(v, v, v, v);
#pragma unroll
    for (int j = 0; j < COP / 4; ++j) reinterpret_cast<float4*>(lut + threadIdx.x * COP)[j] = v4;
    __syncthreads();
  }
  const float* mylut = lut + (threadIdx.x & (COP - 1));
  const int W1 = Wi / 2, W2 = Wi / 4, H1 = Hi / 2, H2 = Hi / 4;
  const int tw = Wi / 8, th = Hi / 8;  // tiles per row / tile rows
  const long long total = (long long)planes * th * tw;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    int tx, ty, pl;
    if (total <= 0xffffffffLL) {  // (wave-uniform) round 4: three 64-bit divisions per tile were ~450 of the
      // thread's ~700 VALU instructions (common.h FastDiv)
      const unsigned rowi = fastdiv((unsigned)t, by_tw);
      tx = (int)((unsigned)t - rowi * (unsigned)tw);
      pl = (int)fastdiv(rowi, by_th);
      ty = (int)(rowi - (unsigned)pl * (unsigned)th);
    } else {
      tx = (int)(t % tw);
      ty = (int)((t / tw) % th);
      pl = (int)(t / ((long long)tw * th));
    }
    constexpr int NC = HWC ? 3 : 1, RD = HWC ? 6 : 2;  // channels per thread, dwords per tile row
    const uint8_t* sp = src + (((size_t)pl * Hi + (size_t)ty * 8) * Wi + (size_t)tx * 8) * NC;
    uint32_t raw[8][RD];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int k = 0; k < RD / 2; ++k) {
        const uint2 w = *reinterpret_cast<const uint2*>(sp + (size_t)r * Wi * NC + 8 * k);
        raw[r][2 * k] = w.x;
        raw[r][2 * k + 1] = w.y;
      }
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
    const int opl = HWC ? pl * 3 + ch : pl;  // the plane of levels 1 / 2 this channel goes to
    float q[4][4];  // level 1: 4 rows x 4 px
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v[2][8];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int px = 0; px < 8; ++px) {
          const int k = px * NC + ch;  // byte of the row segment (compile-time after unrolling)
          v[i][px] = mylut[((raw[2 * r + i][k >> 2] >> ((k & 3) * 8)) & 0xff) * COP];
        }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        q[r][c] = __fadd_rn(__fadd_rn(__fadd_rn(v[0][2 * c], v[0][2 * c + 1]), v[1][2 * c]),
                            v[1][2 * c + 1]) * 0.25f;
      *reinterpret_cast<float4*>(l1 + ((size_t)opl * H1 + (size_t)ty * 4 + r) * W1 + (size_t)tx * 4) =
          make_float4(q[r][0], q[r][1], q[r][2], q[r][3]);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      float o[2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
        o[c] = __fadd_rn(__fadd_rn(__fadd_rn(q[2 * r][2 * c], q[2 * r][2 * c + 1]), q[2 * r + 1][2 * c]),
                         q[2 * r + 1][2 * c + 1]) * 0.25f;
      *reinterpret_cast<float2*>(l2 + ((size_t)opl * H2 + (size_t)ty * 2 + r) * W2 + (size_t)tx * 2) =
          make_float2(o[0], o[1]);
    }
    }  // ch
  }
}

// ------------------------------------------------------------------------------------------------
// geometry: one thread per (aug, box); fp64 internally, fp32 in/out.

struct D3 { double x, y, z; };
__device__ __forceinline__ D3 cross3(D3 a, D3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double norm3(D3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

__device__ __forceinline__ void inv3x3(const double* m, double* o) {
  const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7],
               i = m[8];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double inv = 1.0 / (a * A + b * B + c * C);
  o[0] = A * inv; o[1] = -(b * i - c * h) * inv; o[2] = (b * f - c * e) * inv;
  o[3] = B * inv; o[4] = (a * i - c * g) * inv;  o[5] = -(a * f - c * d) * inv;
  o[6] = C * inv; o[7] = -(a * h - b * g) * inv; o[8] = (a * e - b * d) * inv;
}
__device__ __forceinline__ void matmul3(const double* a, const double* b, double* o) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      o[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
}

// OpenCV rational + tangential + thin-prism model, warping.py:90-107
template <typename F>
__device__ __forceinline__ void distortion_parts(F x, F y, const F* d, F& a, F& b, F& cx, F& cy) {
  const F r2 = x * x + y * y;
  a = (((d[4] * r2 + d[1]) * r2 + d[0]) * r2 + F(1)) / (((d[7] * r2 + d[6]) * r2 + d[5]) * r2 + F(1));
  b = F(2) * (x * d[3] + y * d[2]);
  cx = (d[9] * r2 + d[3] + d[8]) * r2;
  cy = (d[11] * r2 + d[2] + d[10]) * r2;
}

__global__ __launch_bounds__(64) void crop_geometry_kernel(
    const float* __restrict__ boxes, int box_stride, const float* __restrict__ intr,
    const float* __restrict__ dist, const float* __restrict__ up, const int32_t* __restrict__ ids,
    const float* __restrict__ rotflip, const float* __restrict__ aug_scales,
    const float* __restrict__ aug_gammas, int n_box, int n_aug, int res, int aa,
    float* __restrict__ new_k, float* __restrict__ rot_out, float* __restrict__ wp) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_box * n_aug) return;
  const int a = t / n_box, i = t % n_box;  // crops are [aug, box]-major (multiperson_model.py:237-240)

  double K[9], Kinv[9], dc[12];
  bool has_dist = false;
#pragma unroll
  for (int k = 0; k < 9; ++k) K[k] = intr[(size_t)i * 9 + k];
#pragma unroll
  for (int k = 0; k < 12; ++k) { dc[k] = dist[(size_t)i * 12 + k]; has_dist |= (dc[k] != 0.0); }
  inv3x3(K, Kinv);

  // five box points: centre, top-, right-, bottom-, left-mid (multiperson_model.py:325-330)
  const double bx = boxes[(size_t)i * box_stride], by = boxes[(size_t)i * box_stride + 1],
               bw = boxes[(size_t)i * box_stride + 2], bh = boxes[(size_t)i * box_stride + 3];
  const double px[5] = {bx + bw / 2, bx + bw / 2, bx + bw, bx + bw / 2, bx};
  const double py[5] = {by + bh / 2, by, by + bh / 2, by + bh, by + bh / 2};
  double cx[5], cy[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    // [x, y, 1] @ inv(K)^T, first two components (the reference slices [:, :, :2], :334)
    const double ux = Kinv[0] * px[k] + Kinv[1] * py[k] + Kinv[2];
    const double uy = Kinv[3] * px[k] + Kinv[4] * py[k] + Kinv[5];
    double x = ux, y = uy;
    if (has_dist) {  // undistort_points: 5 fixed-point iterations (warping.py:65-73)
      for (int itn = 0; itn < 5; ++itn) {
        double pa, pb, pcx, pcy;
        distortion_parts<double>(x, y, dc, pa, pb, pcx, pcy);
        const double nx = (ux - pcx - x * pb) / pa, ny = (uy - pcy - y * pb) / pa;
        x = nx; y = ny;
      }
    }
    cx[k] = x; cy[k] = y;
  }
  // look-at rotation (ptu3d.py:129-142)
  D3 f{cx[0], cy[0], 1.0};
  const double fn = norm3(f);
  D3 z{f.x / fn, f.y / fn, f.z / fn};
  D3 upv{up[(size_t)i * 3], up[(size_t)i * 3 + 1], up[(size_t)i * 3 + 2]};
  D3 x = cross3(z, upv);
  if (norm3(x) == 0.0) x = D3{z.z, 0.0, -z.x};
  const double xn = norm3(x);
  x = D3{x.x / xn, x.y / xn, x.z / xn};
  D3 y = cross3(z, x);
  double Rn[9] = {x.x, x.y, x.z, y.x, y.y, y.z, z.x, z.y, z.z};

  // side mid-points through K @ R_noaug (multiperson_model.py:343-345)
  double M[9];
  matmul3(K, Rn, M);
  double sx[4], sy[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double qx = M[0] * cx[k + 1] + M[1] * cy[k + 1] + M[2];
    const double qy = M[3] * cx[k + 1] + M[4] * cy[k + 1] + M[5];
    const double qz = M[6] * cx[k + 1] + M[7] * cy[k + 1] + M[8];
    sx[k] = qx / qz; sy[k] = qy / qz;
  }
  const double vertical = hypot(sx[0] - sx[2], sy[0] - sy[2]);
  const double horizontal = hypot(sx[1] - sx[3], sy[1] - sy[3]);
  const double box_scale = (double)res / fmax(vertical, horizontal);
  const float crop_scale_f = (float)((double)aug_scales[a] * (double)(float)box_scale);
  const double s = crop_scale_f;

  // new intrinsics: top-left 2x2 of K scaled, principal point res/2 (:277-286)
  double NK[9] = {K[0] * s, K[1] * s, res / 2.0, K[3] * s, K[4] * s, res / 2.0, 0.0, 0.0, 1.0};
  double RF[9], R[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) RF[k] = rotflip[(size_t)a * 9 + k];
  matmul3(RF, Rn, R);  // R = rotflip_aug @ R_noaug (:287)
  double P[9], Hinv[9];
  matmul3(NK, R, P);
  inv3x3(P, Hinv);  // new_invprojmat = inv(K_new @ R) (:288)
  if (aa > 1) {     // @ corner_aligned_scale_mat(1/aa) (:292-295, warping.py:128-133)
    const double fct = 1.0 / aa, sh = (fct - 1.0) / 2.0;
    double S[9] = {fct, 0, sh, 0, fct, sh, 0, 0, 1}, T[9];
    matmul3(Hinv, S, T);
#pragma unroll
    for (int k = 0; k < 9; ++k) Hinv[k] = T[k];
  }
  // pyramid level = clip(floor(-log2(crop_scale * aa)), 0, 2) (warping.py:20-21, :303)
  const float lv = floorf(-log2f(__fmul_rn(crop_scale_f, (float)aa)));
  const int level = (int)fminf(fmaxf(lv, 0.0f), 2.0f);
  const double fl = 1.0 / (double)(1 << level), shl = (fl - 1.0) / 2.0;
  double SL[9] = {fl, 0, shl, 0, fl, shl, 0, 0, 1}, KL[9];
  matmul3(SL, K, KL);  // corner_aligned_scale_mat(2^-l) @ K (warping.py:15-17)

  const size_t o = (size_t)a * n_box + i;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    new_k[o * 9 + k] = (float)NK[k];
    rot_out[o * 9 + k] = (float)R[k];
    wp[o * MTR_WARP_PARAM_FLOATS + k] = (float)Hinv[k];
    wp[o * MTR_WARP_PARAM_FLOATS + 9 + k] = (float)KL[k];
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) wp[o * MTR_WARP_PARAM_FLOATS + 18 + k] = (float)dc[k];
  wp[o * MTR_WARP_PARAM_FLOATS + 30] = has_dist ? 1.0f : 0.0f;
  wp[o * MTR_WARP_PARAM_FLOATS + 31] = (float)level;
  wp[o * MTR_WARP_PARAM_FLOATS + 32] = (float)ids[i];
  wp[o * MTR_WARP_PARAM_FLOATS + 33] = __fdiv_rn(aug_gammas[a], 2.2f);  // gamma / 2.2 (:319)
  wp[o * MTR_WARP_PARAM_FLOATS + 34] = crop_scale_f;
  wp[o * MTR_WARP_PARAM_FLOATS + 35] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// warp: thread = PX consecutive output pixels of one row, all 3 channels.
//
// v1 replayed every fp32 operation of the reference (IEEE divisions, the [-1,1] normalise /
// un-normalise round trip of grid_sample, libm powf) and cost ~900 VALU instructions per pixel: the
// kernel was ALU-bound at 25 % of its HBM roofline.  v2 keeps the reference's geometry
// (homography, distortion polynomial, K_level, bilinear taps with zero padding, align_corners=True
// pixel coordinates) but
//   * evaluates the sample position directly (the round trip only adds rounding noise the
//     reference itself is subject to: its own fp32-vs-fp64 floor is 1e-5 in linear light);
//   * one IEEE reciprocal per pixel for the perspective divide;
//   * x**g as exp2(g*log2(x)) on v_log_f32 / v_exp_f32 (<= 4 ulp for dark pixels, 1 ulp above
//     0.25; 0 -> 0 exactly);
//   * taps as two 8-byte buffer loads per channel (columns xs, xs+1 of rows ys, ys+1 with
//     xs = clamp(x0, 0, W-2)), zero padding folded into four per-pixel weights; the buffer
//     descriptor (one per crop, wave-uniform) range-checks the address, so no per-tap branches.

struct LevelDims { int H[3], W[3]; };

__device__ __forceinline__ float tap(const float* __restrict__ plane, int x, int y, int W, int H) {
  // grid_sample(padding_mode='zeros'): a tap outside [0,W-1]x[0,H-1] contributes 0
  return (x >= 0 && x < W && y >= 0 && y < H) ? plane[(size_t)y * W + x] : 0.0f;
}

template <typename OutT> __device__ __forceinline__ OutT from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ __hip_bfloat16 from_f32<__hip_bfloat16>(float v) {
  return __float2bfloat16(v);
}

__device__ __forceinline__ float fast_pow_unit(float x, float g) {
  // x in [0, 1]: log2(0) = -inf -> exp2(-inf) = 0
  return __builtin_amdgcn_exp2f(g * __builtin_amdgcn_logf(x));
}

// weights of the loaded pair (columns s, s+1) for the wanted taps (i, i+1) with fractions
// (t0 for i, t1 for i+1); n = extent.  Taps outside [0, n-1] contribute zero.
__device__ __forceinline__ void pair_weights(int i, int s, int n, float t0, float t1, float& w_first,
                                             float& w_second) {
  w_first = 0.0f;
  w_second = 0.0f;
  if (i == s) { w_first = t0; w_second = t1; }       // both taps inside
  else if (i == -1) { w_first = t1; }                // only tap i+1 = 0 is inside
  else if (i == n - 1) { w_second = t0; }            // only tap i = n-1 is inside (s = n-2)
}

// (px_stride 1: a plane of [N,3,H,W]; 3: one channel of an interleaved [N,H,W,3] frame)
__device__ __forceinline__ float tap_u8(const uint8_t* __restrict__ plane, const float* lut, int x,
                                        int y, int W, int H, int px_stride = 1) {
  return (x >= 0 && x < W && y >= 0 && y < H) ? lut[plane[((size_t)y * W + x) * px_stride]] : 0.0f;
}

// L0U8: level 0 is the uint8 frame itself (l0 is then a uint8 pointer) decoded through a copy of
// the gamma LUT in LDS.  One 8-byte load from the 4-byte-aligned address below the tap still
// yields both x-taps of a row: bytes (off&3) and (off&3)+1 of the 64-bit word.
// L0 = 2: the uint8 frame is interleaved [N,H,W,3]; the two x-taps of a channel are then bytes
// (off & 3) and (off & 3) + 3 of the same 8-byte window.
template <typename OutT, int AA, int PX, int L0>
__global__ __launch_bounds__(256) void warp_crops_kernel(
    const void* __restrict__ l0_any, const float* __restrict__ l1, const float* __restrict__ l2,
    const float* __restrict__ lut_g, LevelDims dims, unsigned u8_bytes,
    const float* __restrict__ wp_all, int n_crops, int res, int nhwc, OutT* __restrict__ out) {
  constexpr bool L0U8 = L0 != 0, HWC = L0 == 2;
  __shared__ float lut[L0U8 ? 256 : 1];
  if (L0U8) {
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();
  }
  const float* __restrict__ l0 = (const float*)l0_any;
  // ---- XCD-aware map (block id b runs on XCD b % 8): all row tiles of a crop share one XCD, so
  // the crop's source footprint is fetched into ONE L2 instead of eight.
  // a wave covers a 64-px x 4-row output tile (16 lanes x PX pixels per row); a block = 4 such
  // tiles stacked (64 x 16): under the +-25 degree TTA rotations a 256 x 1 row per wave walked
  // ~100 source rows, this keeps a wave's source footprint within ~30 rows
  const int row_tiles = (res + 15) / 16;
  const int x_tiles = (res + 16 * PX - 1) / (16 * PX);
  const int per_crop = row_tiles * x_tiles;
  const int id = blockIdx.x;
  const int crop = (id / (8 * per_crop)) * 8 + (id % 8);
  if (crop >= n_crops) return;
  const int tile = (id / 8) % per_crop;
  const int ty = tile / x_tiles, tx = tile - ty * x_tiles;

  const float* __restrict__ wp = wp_all + (size_t)crop * MTR_WARP_PARAM_FLOATS;
  // per-crop constants are wave-uniform -> scalar loads
  const float h0 = wp[0], h1 = wp[1], h2 = wp[2], h3 = wp[3], h4 = wp[4], h5 = wp[5], h6 = wp[6],
              h7 = wp[7], h8 = wp[8];
  const float k0 = wp[9], k1 = wp[10], k2 = wp[11], k3 = wp[12], k4 = wp[13], k5 = wp[14];
  const bool has_dist = wp[30] != 0.0f;
  const int level = (int)wp[31];
  const int img = (int)wp[32];
  const float gexp = wp[33];
  const int W = dims.W[level], H = dims.H[level];
  const bool bytes0 = L0U8 && level == 0;  // wave-uniform
  const float* __restrict__ lvl = level == 0 ? l0 : (level == 1 ? l1 : l2);
  const float* __restrict__ planes = lvl + (size_t)img * 3 * H * W;
  const uint8_t* __restrict__ planes_u8 = (const uint8_t*)l0_any + (size_t)img * 3 * H * W;
  const int plane_elems = H * W;
  // uint8 path: the descriptor spans the WHOLE frame tensor (its base is 4-byte aligned; an
  // image's own base is not when 3*H*W is odd) and the image offset joins the byte offset
  const int img_off = bytes0 ? img * 3 * plane_elems : 0;
  const buffer_rsrc_t rsrc =
      bytes0 ? make_rsrc(uniform_ptr((const uint8_t*)l0_any), (unsigned)u8_bytes)
             : make_rsrc(uniform_ptr(planes), (unsigned)(3 * plane_elems) * 4u);
  const bool tiny = W < 2 || H < 2;  // degenerate pyramid levels: per-tap path

  const int lane = threadIdx.x & 63;
  const int u0 = (tx * 16 + (lane & 15)) * PX;
  const int v = ty * 16 + (threadIdx.x >> 6) * 4 + (lane >> 4);
  if (v >= res || u0 >= res) return;

  float acc[PX][3];
#pragma unroll
  for (int p = 0; p < PX; ++p) acc[p][0] = acc[p][1] = acc[p][2] = 0.0f;

#pragma unroll
  for (int p = 0; p < PX; ++p) {
#pragma unroll
    for (int sj = 0; sj < AA; ++sj) {
#pragma unroll
      for (int si = 0; si < AA; ++si) {
        const float U = (float)((u0 + p) * AA + si), V = (float)(v * AA + sj);
        // old = Hinv @ [U, V, 1] (warping.py:45-46); project (ptu3d.py:145-146)
        const float ox = fmaf(h0, U, fmaf(h1, V, h2));
        const float oy = fmaf(h3, U, fmaf(h4, V, h5));
        const float oz = fmaf(h6, U, fmaf(h7, V, h8));
        const float inv = __fdiv_rn(1.0f, oz);
        float nx = ox * inv, ny = oy * inv;
        if (has_dist) {  // distort_points (warping.py:57-62)
          float pa, pb, pcx, pcy;
          distortion_parts<float>(nx, ny, wp + 18, pa, pb, pcx, pcy);
          const float sc = pa + pb;
          nx = fmaf(nx, sc, pcx);
          ny = fmaf(ny, sc, pcy);
        }
        // pixel coordinates in the chosen level: (K_lvl @ [nx, ny, 1])[:2] (warping.py:49);
        // align_corners=True => these ARE the sample coordinates
        const float ix = fmaf(k0, nx, fmaf(k1, ny, k2));
        const float iy = fmaf(k3, nx, fmaf(k4, ny, k5));
        // taps entirely outside the frame (or non-finite coordinates) contribute nothing
        const bool sane = (ix > -1.0f) && (iy > -1.0f) && (ix < (float)W) && (iy < (float)H);
        if (!sane) continue;
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float tx1 = ix - fx0, tx0 = 1.0f - tx1;
        const float ty1 = iy - fy0, ty0 = 1.0f - ty1;
        if (!tiny) {
          const int xs = min(max(x0, 0), W - 2), ys = min(max(y0, 0), H - 2);
          float wl, wr, wt, wb;
          pair_weights(x0, xs, W, tx0, tx1, wl, wr);
          pair_weights(y0, ys, H, ty0, ty1, wt, wb);
          const float w00 = wl * wt, w01 = wr * wt, w10 = wl * wb, w11 = wr * wb;
          if (!bytes0) {
            const int off = (ys * W + xs) * 4;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const auto top = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, c * plane_elems * 4, 0);
              const auto bot = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + W * 4, c * plane_elems * 4, 0);
              struct F2 { float a, b; };
              const F2 t = __builtin_bit_cast(F2, top), b2 = __builtin_bit_cast(F2, bot);
              acc[p][c] += fmaf(b2.b, w11, fmaf(b2.a, w10, fmaf(t.b, w01, t.a * w00)));
            }
          } else {
            unsigned long long tw_keep = 0, bw_keep = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              // byte offsets of the top-left tap and of the one below it
              const int ot = HWC ? img_off + (ys * W + xs) * 3 + c : img_off + c * plane_elems + ys * W + xs;
              const int ob = ot + (HWC ? 3 * W : W);
              constexpr int RS = HWC ? 24 : 8;  // bit position of the right-hand tap in the window
              unsigned long long tw, bw;
              if ((MTR_WARP_ABLATE & 16) && c > 0) {  // timing probe: one gather pair serves all channels
                tw = tw_keep; bw = bw_keep;
              } else if (MTR_WARP_ABLATE & 1) {
                tw = (unsigned long long)(unsigned)ot * 2654435761ull;
                bw = (unsigned long long)(unsigned)ob * 2654435761ull;
              } else {
                const auto top = __builtin_amdgcn_raw_buffer_load_b64(rsrc, ot & ~3, 0, 0);
                const auto bot = __builtin_amdgcn_raw_buffer_load_b64(rsrc, ob & ~3, 0, 0);
                tw = __builtin_bit_cast(unsigned long long, top) >> ((ot & 3) * 8);
                bw = __builtin_bit_cast(unsigned long long, bot) >> ((ob & 3) * 8);
              }
              if (MTR_WARP_ABLATE & 16) { tw_keep = tw; bw_keep = bw; }
              float ta, tb, ba, bb;
              if (MTR_WARP_ABLATE & 2) {
                ta = (float)(tw & 0xff); tb = (float)((tw >> RS) & 0xff);
                ba = (float)(bw & 0xff); bb = (float)((bw >> RS) & 0xff);
              } else {
                ta = lut[tw & 0xff]; tb = lut[(tw >> RS) & 0xff];
                ba = lut[bw & 0xff]; bb = lut[(bw >> RS) & 0xff];
              }
              acc[p][c] += fmaf(bb, w11, fmaf(ba, w10, fmaf(tb, w01, ta * w00)));
            }
          }
        } else if (bytes0) {
          const float wnw = tx0 * ty0, wne = tx1 * ty0, wsw = tx0 * ty1, wse = tx1 * ty1;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const uint8_t* __restrict__ pl = planes_u8 + (HWC ? (size_t)c : (size_t)c * plane_elems);
            constexpr int PS = HWC ? 3 : 1;
            const float nw = tap_u8(pl, lut, x0, y0, W, H, PS), ne = tap_u8(pl, lut, x0 + 1, y0, W, H, PS);
            const float sw = tap_u8(pl, lut, x0, y0 + 1, W, H, PS), se = tap_u8(pl, lut, x0 + 1, y0 + 1, W, H, PS);
            acc[p][c] += fmaf(se, wse, fmaf(sw, wsw, fmaf(ne, wne, nw * wnw)));
          }
        } else {
          const float wnw = tx0 * ty0, wne = tx1 * ty0, wsw = tx0 * ty1, wse = tx1 * ty1;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float* __restrict__ pl = planes + (size_t)c * plane_elems;
            const float nw = tap(pl, x0, y0, W, H), ne = tap(pl, x0 + 1, y0, W, H);
            const float sw = tap(pl, x0, y0 + 1, W, H), se = tap(pl, x0 + 1, y0 + 1, W, H);
            acc[p][c] += fmaf(se, wse, fmaf(sw, wsw, fmaf(ne, wne, nw * wnw)));
          }
        }
      }
    }
  }

  // antialias average and the per-crop gamma (multiperson_model.py:308-319)
  float res_v[PX][3];
#pragma unroll
  for (int p = 0; p < PX; ++p)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float val = acc[p][c];
      if (AA > 1) val = val * (1.0f / (AA * AA));
      res_v[p][c] = (gexp == 1.0f || (MTR_WARP_ABLATE & 4)) ? val : fast_pow_unit(val, gexp);
    }

  const bool full = (u0 + PX <= res) && (res % PX == 0);
  if ((MTR_WARP_ABLATE & 8) && (res_v[0][0] != 12345.0f || lane != 0)) return;
  if (!nhwc) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      OutT* dst = out + (((size_t)crop * 3 + c) * res + v) * res + u0;
      if (full && PX % 4 == 0 && sizeof(OutT) == 4) {
#pragma unroll
        for (int q = 0; q < PX; q += 4)
          *reinterpret_cast<float4*>(dst + q) =
              make_float4(res_v[q][c], res_v[q + 1][c], res_v[q + 2][c], res_v[q + 3][c]);
      } else if (full && PX % 4 == 0 && sizeof(OutT) == 2) {
#pragma unroll
        for (int q = 0; q < PX; q += 4) {
          OutT tmp[4] = {from_f32<OutT>(res_v[q][c]), from_f32<OutT>(res_v[q + 1][c]),
                         from_f32<OutT>(res_v[q + 2][c]), from_f32<OutT>(res_v[q + 3][c])};
          *reinterpret_cast<uint2*>(dst + q) = *reinterpret_cast<uint2*>(tmp);
        }
      } else if (full && PX == 2 && sizeof(OutT) == 4) {
        *reinterpret_cast<float2*>(dst) = make_float2(res_v[0][c], res_v[1][c]);
      } else {
#pragma unroll
        for (int p = 0; p < PX; ++p)
          if (u0 + p < res) dst[p] = from_f32<OutT>(res_v[p][c]);
      }
    }
  } else {
    OutT* dst = out + (((size_t)crop * res + v) * res + u0) * 3;
#pragma unroll
    for (int p = 0; p < PX; ++p)
      if (u0 + p < res) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[p * 3 + c] = from_f32<OutT>(res_v[p][c]);
      }
  }
}

// ---- row-walking variant of the sampler ---------------------------------------------------------
// Same arithmetic per sample as warp_crops_kernel (1 / z by v_rcp_f32 + a Newton step instead of
// the IEEE division sequence: the only difference), different schedule: a lane owns ONE output
// column and walks ROWS rows of it (a wave = LX columns x 64 / LX rows per step); the taps of
// sample s+1 are requested before sample s is finished, and a finished pixel is stored at once
// (LX lanes x 4 B = 128 contiguous bytes per row and channel).  Stores therefore leave a wave as a
// steady stream instead of one burst at its end, and
// -- gfx9 counts loads and stores in the same in-order vmcnt -- the next taps are always OLDER than
// the previous pixel's stores, so waiting for taps never waits for a store.
// ---- counted waits for the row-walking sampler (MTR_WARP_ASM) ---------------------------------------------------
using v4i_sgpr = __attribute__((ext_vector_type(4))) int;
// the descriptor words of make_rsrc as four SGPRs an asm operand can name
__device__ __forceinline__ v4i_sgpr make_rsrc_words(const void* wave_uniform_base, unsigned bytes) {
  const unsigned long long v = (unsigned long long)wave_uniform_base;
  v4i_sgpr r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(v >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
// 8 bytes per lane through a buffer descriptor, asynchronous: the caller waits with tap_wait<N> before it reads r
__device__ __forceinline__ unsigned long long tap_load_asm(v4i_sgpr rsrc, int voff, int soff) {
  unsigned long long r;
  asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=&v"(r) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  return r;
}
// Vector-memory operations the row walk issues between the LAST tap load of sample s and the point where sample s is
// finished (the loads of the samples requested ahead, 6 each; the stores of the pixels completed meanwhile, 3 each):
// a replay of the loop below.  gfx9 retires loads and stores through one in-order counter.
constexpr int warp_ops_after(int s, int NS, int PD, int AA2) {
  int ops = 0;
  bool seen = false;
  for (int q = 0; q < PD && q < NS; ++q) {  // the prologue's requests
    if (seen) ops += 6;
    if (q == s) seen = true;
  }
  for (int i = 0; i < NS; ++i) {
    if (i + PD < NS) {
      if (seen) ops += 6;
      if (i + PD == s) seen = true;
    }
    if (i == s) return ops;
    if ((i + 1) % AA2 == 0 && seen) ops += 3;  // (the pixel's three stores: always issued, see the store below)
  }
  return ops;
}

struct TapSet {
  unsigned long long raw[6];  // [channel][top, bottom]: the two x-taps of a row (f32 pair / byte window)
  float w00, w01, w10, w11;
  int off;                    // element (f32 levels) or byte (uint8 level 0) offset of the top-left tap
};

// L0: representation of level 0 -- 0 = f32 planes, 1 = the uint8 frame [N,3,H,W], 2 = the uint8 frame
// with interleaved channels [N,H,W,3] (what decoders and numpy hand over): there the two x-taps of a
// row for ALL three channels are six consecutive bytes, so one 12-byte load from the aligned address
// below them replaces three 8-byte ones -- two gathers per sample instead of six.
// PERSIST (round 6, developer builds: -DMTR_WARP_PERSIST=<workgroups per CU>): a FIXED grid of workgroups, each
// walking a contiguous run of (crop, tile) items of its XCD crop-major -- the LUT is filled once per workgroup and
// a crop's 36 warp-row scalars, descriptors and level choice are set up once per run of its tiles instead of once
// per 32 x 32 tile.  Same per-sample arithmetic, same bits.
template <typename OutT, int AA, int ROWS, int L0, bool PERSIST = false>
__global__ __launch_bounds__(64 * MTR_WARP_WAVES) void warp_rows_kernel(
    const void* __restrict__ l0_any, const float* __restrict__ l1, const float* __restrict__ l2,
    const float* __restrict__ lut_g, LevelDims dims, unsigned u8_bytes,
    const float* __restrict__ wp_all, int n_crops, int res, int nhwc, OutT* __restrict__ out) {
  constexpr bool L0U8 = L0 != 0, HWC = L0 == 2;
  constexpr int NWV = MTR_WARP_WAVES;
  __shared__ float lut[L0U8 ? 256 : 1];
  if (L0U8) {
#pragma unroll
    for (int i = threadIdx.x; i < 256; i += 64 * NWV) lut[i] = lut_g[i];
    __syncthreads();
  }
  const float* __restrict__ l0 = (const float*)l0_any;
  // block = LX columns x 4*ROWS*RI rows (wave w owns the w-th band of ROWS*RI rows and walks it RI
  // rows at a time); all tiles of a crop on one XCD, as in warp_crops_kernel
  constexpr int LX = MTR_WARP_LX, RI = 64 / LX;  // a wave iteration covers LX columns x RI rows
  const int row_tiles = (res + NWV * ROWS * RI - 1) / (NWV * ROWS * RI);
  const int x_tiles = (res + LX - 1) / LX;
  const int per_crop = row_tiles * x_tiles;
  const int id = blockIdx.x;
  // items of this workgroup: one (its own tile), or -- PERSIST -- the slot-th run of the items of XCD id % 8, whose
  // crops are 8 k + id % 8 and whose items are (k, tile) in crop-major order
  int item = 0, item_end = 1;
  if constexpr (PERSIST) {
    const int slots = gridDim.x >> 3, slot = id >> 3;
    const int total = ((n_crops + 7) >> 3) * per_crop;
    const int run = (total + slots - 1) / slots;
    item = slot * run;
    item_end = min(total, item + run);
  }
  for (; item < item_end; ++item) {
  const int crop = PERSIST ? (item / per_crop) * 8 + (id & 7) : (id / (8 * per_crop)) * 8 + (id % 8);
  if (crop >= n_crops) continue;
  const int tile = PERSIST ? item % per_crop : (id / 8) % per_crop;
  const int ty = tile / x_tiles, tx = tile - ty * x_tiles;

  const float* __restrict__ wp = wp_all + (size_t)crop * MTR_WARP_PARAM_FLOATS;
  const float h0 = wp[0], h1 = wp[1], h2 = wp[2], h3 = wp[3], h4 = wp[4], h5 = wp[5], h6 = wp[6],
              h7 = wp[7], h8 = wp[8];
  const float k0 = wp[9], k1 = wp[10], k2 = wp[11], k3 = wp[12], k4 = wp[13], k5 = wp[14];
  const bool has_dist = wp[30] != 0.0f;
  // (wave-uniform by construction; said explicitly so that every descriptor / scalar offset below
  //  stays in SGPRs across the unrolled sample pipeline)
  const int level = __builtin_amdgcn_readfirstlane((int)wp[31]);
  const int img = __builtin_amdgcn_readfirstlane((int)wp[32]);
  const float gexp = wp[33];
  const int W = __builtin_amdgcn_readfirstlane(dims.W[level]);
  const int H = __builtin_amdgcn_readfirstlane(dims.H[level]);
  const bool bytes0 = L0U8 && level == 0;  // wave-uniform
  const float* __restrict__ lvl = level == 0 ? l0 : (level == 1 ? l1 : l2);
  const float* __restrict__ planes = lvl + (size_t)img * 3 * H * W;
  const int plane_elems = __builtin_amdgcn_readfirstlane(H * W);
  const int img_off = __builtin_amdgcn_readfirstlane(bytes0 ? img * 3 * plane_elems : 0);
  // ONE request path for both kinds of level: byte offset of the top-left tap = ((ys*W + xs) << sh)
  // + base, six loads at (offset & ~3) through one descriptor (uint8 level 0: the whole frame
  // tensor, see warp_crops_kernel; f32 levels: the image's three planes)
  const int sh = bytes0 ? 0 : 2;
  const int plane_bytes = __builtin_amdgcn_readfirstlane(plane_elems << sh);
  const int row_bytes = __builtin_amdgcn_readfirstlane(HWC && bytes0 ? 3 * W : W << sh);
  const buffer_rsrc_t rsrc =
      bytes0 ? make_rsrc(uniform_ptr((const uint8_t*)l0_any), (unsigned)u8_bytes)
             : make_rsrc(uniform_ptr(planes), (unsigned)(3 * plane_elems) * 4u);

  const int x = tx * LX + (threadIdx.x & (LX - 1));
  const int v_first = (ty * NWV + (threadIdx.x >> 6)) * ROWS * RI + ((threadIdx.x & 63) / LX);
  if (x >= res || v_first >= res) continue;
#if !MTR_WARP_LEAN
  const float fW = (float)W, fH = (float)H;
#endif
  const bool same_shift = ((row_bytes | plane_bytes) & 3) == 0;
  // crop stores through a descriptor over this crop: 32-bit offsets, the channel pitch in an SGPR
  const buffer_rsrc_t orsrc = make_rsrc(uniform_ptr(out + (size_t)crop * 3 * res * res),
                                        (unsigned)(3 * res * res) * (unsigned)sizeof(OutT));
  const int chan_bytes = __builtin_amdgcn_readfirstlane(res * res * (int)sizeof(OutT));

  // The sample pipeline, instantiated per kind of source: IL = interleaved uint8 level 0 (only in the
  // L0 == 2 kernel, whose levels 1 and 2 are f32 planes like everyone's: the split is made ONCE per
  // wave, below, so that each instance stays straight-line code).
  auto run = [&](auto il_tag, auto asm_tag) {
  constexpr bool IL = decltype(il_tag)::value;
  // ASMP: tap loads from inline asm + counted waits (MTR_WARP_ASM; the pitches are multiples of 4: one address register,
  // the plane / row offsets in six scalars).  Straight-line code between a load and its wait: no branch may separate them
  // (the compiler would copy the tied registers in front of the wait).
  constexpr bool ASMP = decltype(asm_tag)::value;
  v4i_sgpr rsrc_w = {0, 0, 0, 0};
  int soffs[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (ASMP) {
    rsrc_w = bytes0 ? make_rsrc_words(uniform_ptr((const uint8_t*)l0_any), (unsigned)u8_bytes)
                    : make_rsrc_words(uniform_ptr(planes), (unsigned)(3 * plane_elems) * 4u);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      soffs[2 * c] = __builtin_amdgcn_readfirstlane(c * plane_bytes);
      soffs[2 * c + 1] = __builtin_amdgcn_readfirstlane(c * plane_bytes + row_bytes);
    }
  }
  // sample s of this lane: row s / (AA*AA), sub-sample (sj, si) in the reference's loop order
  auto request = [&](int s) -> TapSet {
    const int r = s / (AA * AA), sj = (s / AA) % AA, si = s % AA;
    const float U = (float)(x * AA + si), V = (float)((v_first + r * RI) * AA + sj);
    const float ox = fmaf(h0, U, fmaf(h1, V, h2));
    const float oy = fmaf(h3, U, fmaf(h4, V, h5));
    const float oz = fmaf(h6, U, fmaf(h7, V, h8));
#if MTR_WARP_RCP
    float inv = __builtin_amdgcn_rcpf(oz);  // 1 ulp, then one Newton step: <= 1 ulp of 1/oz
    inv = fmaf(fmaf(-oz, inv, 1.0f), inv, inv);
#else
    const float inv = __fdiv_rn(1.0f, oz);
#endif
    float nx = ox * inv, ny = oy * inv;
#if MTR_WARP_RCP == 2
    // one residual step each: q + (ox - q oz) / oz -- the quotient the reference's IEEE division gives
    // (project, ptu3d.py:124-126) in all but a few last-bit cases, for two FMAs per coordinate
    nx = fmaf(fmaf(-nx, oz, ox), inv, nx);
    ny = fmaf(fmaf(-ny, oz, oy), inv, ny);
#endif
    if (has_dist) {
      float pa, pb, pcx, pcy;
      distortion_parts<float>(nx, ny, wp + 18, pa, pb, pcx, pcy);
      const float sc = pa + pb;
      nx = fmaf(nx, sc, pcx);
      ny = fmaf(ny, sc, pcy);
    }
    const float ix = fmaf(k0, nx, fmaf(k1, ny, k2));
    const float iy = fmaf(k3, nx, fmaf(k4, ny, k5));
    TapSet t;
#if MTR_WARP_LEAN
    // the loaded pair is (xs, xs + 1) with xs clamped into the frame; d = ix - xs.  d in [0, 1): both
    // taps, weights (1 - d, d) -- the reference's (tx0, tx1), d = ix - floor(ix) exactly; d in [-1, 0):
    // only the right tap of (x0, x0 + 1) = (-1, 0) exists, weight 1 + d; d in [1, 2): only the left tap
    // of (W - 1, W), weight 2 - d; anything else (beyond the zero padding, inf, NaN): 0 and 0.
    // v_cvt_i32_f32 saturates and maps NaN to 0; fmaxf / fminf return the non-NaN operand.
    const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
    const int xs = min(max(x0, 0), W - 2), ys = min(max(y0, 0), H - 2);
    const float dxf = ix - (float)xs, dyf = iy - (float)ys;
    const float wl = fmaxf(1.0f - fabsf(dxf), 0.0f), wr = fmaxf(fminf(dxf, 2.0f - dxf), 0.0f);
    const float wt = fmaxf(1.0f - fabsf(dyf), 0.0f), wb = fmaxf(fminf(dyf, 2.0f - dyf), 0.0f);
#else
    const bool sane = (ix > -1.0f) && (iy > -1.0f) && (ix < fW) && (iy < fH);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = sane ? (int)fx0 : 0, y0 = sane ? (int)fy0 : 0;
    const float tx1 = ix - fx0, tx0 = 1.0f - tx1;
    const float ty1 = iy - fy0, ty0 = 1.0f - ty1;
    const int xs = min(max(x0, 0), W - 2), ys = min(max(y0, 0), H - 2);
    // pair_weights as selects: the loaded pair is (xs, xs+1), the wanted taps (x0, x0+1); inside
    // the sane range x0 - xs is 0 (both taps loaded), -1 (only tap x0+1 = 0) or +1 (only tap x0 = W-1)
    const int dx = x0 - xs, dy = y0 - ys;
    float wl = dx == 0 ? tx0 : (dx < 0 ? tx1 : 0.0f), wr = dx == 0 ? tx1 : (dx > 0 ? tx0 : 0.0f);
    const float wt = dy == 0 ? ty0 : (dy < 0 ? ty1 : 0.0f), wb = dy == 0 ? ty1 : (dy > 0 ? ty0 : 0.0f);
    if (!sane) wl = wr = 0.0f;  // (also non-finite coordinates) the sample contributes nothing
#endif
    t.w00 = wl * wt; t.w01 = wr * wt; t.w10 = wl * wb; t.w11 = wr * wb;
    if constexpr (IL) {
      // bytes off .. off + 5 = (R, G, B) of texel xs and of texel xs + 1; the 12 bytes from (off & ~3) hold them
      t.off = (__mul24(ys, W) + xs) * 3 + img_off;
      const int ob = t.off + row_bytes;
      const auto top = __builtin_amdgcn_raw_buffer_load_b96(rsrc, t.off & ~3, 0, 0);
      const auto bot = __builtin_amdgcn_raw_buffer_load_b96(rsrc, ob & ~3, 0, 0);
      t.raw[0] = (unsigned long long)top[0] | ((unsigned long long)top[1] << 32);
      t.raw[1] = top[2];
      t.raw[2] = (unsigned long long)bot[0] | ((unsigned long long)bot[1] << 32);
      t.raw[3] = bot[2];
      return t;
    }
    t.off = ((__mul24(ys, W) + xs) << sh) + img_off;  // (full-rate 24-bit multiply: ys, W < 2^24)
    if constexpr (ASMP) {
      const int base = t.off & ~3;
#pragma unroll
      for (int k = 0; k < 6; ++k) t.raw[k] = tap_load_asm(rsrc_w, base, soffs[k]);
      return t;
    }
    if (MTR_WARP_LEAN && same_shift) {
      // pitches that are multiples of 4: (off + c * plane + r * row) & ~3 = (off & ~3) + c * plane + r * row,
      // i.e. ONE vector address, the rest in the loads' scalar offset
      const int base = t.off & ~3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (MTR_WARP_ABLATE & 1) {  // (timing ablation: no tap loads)
          t.raw[2 * c] = t.raw[2 * c + 1] = (unsigned long long)(unsigned)base * 0x0101010101ull + c;
          continue;
        }
        if ((MTR_WARP_ABLATE & 16) && c > 0) {  // (timing ablation: ONE gather pair serves the three channels)
          t.raw[2 * c] = t.raw[0] + c;
          t.raw[2 * c + 1] = t.raw[1] + c;
          continue;
        }
        t.raw[2 * c] = __builtin_bit_cast(unsigned long long,
            __builtin_amdgcn_raw_buffer_load_b64(rsrc, base, c * plane_bytes, 0));
        t.raw[2 * c + 1] = __builtin_bit_cast(unsigned long long,
            __builtin_amdgcn_raw_buffer_load_b64(rsrc, base, c * plane_bytes + row_bytes, 0));
      }
      return t;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int ot = t.off + c * plane_bytes, ob = ot + row_bytes;
      t.raw[2 * c] = __builtin_bit_cast(unsigned long long,
          __builtin_amdgcn_raw_buffer_load_b64(rsrc, ot & ~3, 0, 0));
      t.raw[2 * c + 1] = __builtin_bit_cast(unsigned long long,
          __builtin_amdgcn_raw_buffer_load_b64(rsrc, ob & ~3, 0, 0));
    }
    return t;
  };

  auto finish = [&](const TapSet& t, float* acc) {
    struct F2 { float a, b; };
    if constexpr (IL) {
      const unsigned at = (unsigned)t.off & 3u, ab = (unsigned)(t.off + row_bytes) & 3u;
      // bytes 0..3 / 4..7 of the window that starts at the top-left tap: R0 G0 B0 R1 / G1 B1 . .
      const unsigned t_lo = __builtin_amdgcn_alignbyte((unsigned)(t.raw[0] >> 32), (unsigned)t.raw[0], at);
      const unsigned t_hi = __builtin_amdgcn_alignbyte((unsigned)t.raw[1], (unsigned)(t.raw[0] >> 32), at);
      const unsigned b_lo = __builtin_amdgcn_alignbyte((unsigned)(t.raw[2] >> 32), (unsigned)t.raw[2], ab);
      const unsigned b_hi = __builtin_amdgcn_alignbyte((unsigned)t.raw[3], (unsigned)(t.raw[2] >> 32), ab);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float ta = lut[(t_lo >> (8 * c)) & 0xff], ba = lut[(b_lo >> (8 * c)) & 0xff];
        const float tb = lut[c == 0 ? t_lo >> 24 : (t_hi >> (8 * (c - 1))) & 0xff];
        const float bb = lut[c == 0 ? b_lo >> 24 : (b_hi >> (8 * (c - 1))) & 0xff];
        acc[c] += fmaf(bb, t.w11, fmaf(ba, t.w10, fmaf(tb, t.w01, ta * t.w00)));
      }
    } else if (!bytes0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const F2 tp = __builtin_bit_cast(F2, t.raw[2 * c]), bt = __builtin_bit_cast(F2, t.raw[2 * c + 1]);
        acc[c] += fmaf(bt.b, t.w11, fmaf(bt.a, t.w10, fmaf(tp.b, t.w01, tp.a * t.w00)));
      }
    } else if (ASMP || (MTR_WARP_LEAN && same_shift)) {
      // the six byte windows of the sample share one alignment: bytes (off & 3), (off & 3) + 1 of each
      // 8-byte word = bytes 0, 1 of v_alignbyte(high dword, low dword, off & 3)
      const unsigned al = (unsigned)t.off & 3u;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const unsigned tw = __builtin_amdgcn_alignbyte((unsigned)(t.raw[2 * c] >> 32), (unsigned)t.raw[2 * c], al);
        const unsigned bw = __builtin_amdgcn_alignbyte((unsigned)(t.raw[2 * c + 1] >> 32), (unsigned)t.raw[2 * c + 1], al);
        float ta, tb, ba, bb;
        if (MTR_WARP_ABLATE & 2) {  // (timing ablation: no LUT reads)
          ta = (float)(tw & 0xff); tb = (float)((tw >> 8) & 0xff);
          ba = (float)(bw & 0xff); bb = (float)((bw >> 8) & 0xff);
        } else {
          ta = lut[tw & 0xff]; tb = lut[(tw >> 8) & 0xff];
          ba = lut[bw & 0xff]; bb = lut[(bw >> 8) & 0xff];
        }
        acc[c] += fmaf(bb, t.w11, fmaf(ba, t.w10, fmaf(tb, t.w01, ta * t.w00)));
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        // (when the row and plane pitches are multiples of 4 -- wave-uniform -- the six byte
        //  windows of a sample share one shift)
        const int ot = same_shift ? t.off : t.off + c * plane_bytes, ob = same_shift ? t.off : ot + row_bytes;
        const unsigned long long tw = t.raw[2 * c] >> ((ot & 3) * 8);
        const unsigned long long bw = t.raw[2 * c + 1] >> ((ob & 3) * 8);
        const float ta = lut[tw & 0xff], tb = lut[(tw >> 8) & 0xff];
        const float ba = lut[bw & 0xff], bb = lut[(bw >> 8) & 0xff];
        acc[c] += fmaf(bb, t.w11, fmaf(ba, t.w10, fmaf(tb, t.w01, ta * t.w00)));
      }
    }
  };

  constexpr int NS = ROWS * AA * AA;
  float acc[3] = {0.0f, 0.0f, 0.0f};
  // samples s+1 .. s+PD are in flight while sample s is finished (slot = s % (PD+1), all indices
  // compile-time after unrolling)
  constexpr int PD = MTR_WARP_PREFETCH < NS ? MTR_WARP_PREFETCH : NS - 1;
  TapSet ring[PD + 1];
#pragma unroll
  for (int s = 0; s < PD; ++s) ring[s] = request(s);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (s + PD < NS) ring[(s + PD) % (PD + 1)] = request(s + PD);
    if constexpr (ASMP) {
      // the six loads of sample s have landed once at most [what was issued behind them] operations are outstanding
      TapSet& t = ring[s % (PD + 1)];
      asm volatile("s_waitcnt vmcnt(%6)"
                   : "+v"(t.raw[0]), "+v"(t.raw[1]), "+v"(t.raw[2]), "+v"(t.raw[3]), "+v"(t.raw[4]), "+v"(t.raw[5])
                   : "n"(warp_ops_after(s, NS, PD, AA * AA))
                   : "memory");
    }
    finish(ring[s % (PD + 1)], acc);
    if ((s + 1) % (AA * AA) == 0) {  // the pixel of row r is complete
      const int v = v_first + (s / (AA * AA)) * RI;
      float px[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float val = acc[c];
        if (AA > 1) val = val * (1.0f / (AA * AA));
        px[c] = (gexp == 1.0f || (MTR_WARP_ABLATE & 4)) ? val : fast_pow_unit(val, gexp);
        acc[c] = 0.0f;
      }
      if ((MTR_WARP_ABLATE & 8) && px[0] != 12345.0f) continue;  // (timing ablation: no stores)
      // (ASMP: the three stores are ALWAYS issued -- the wait counts above include them -- and a row below the crop
      //  aims past the descriptor's range, where the hardware drops the store)
      if (ASMP || v < res) {
        const int pix = v * res + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const OutT o = from_f32<OutT>(px[c]);
          int voff = (nhwc ? pix * 3 + c : pix) * (int)sizeof(OutT);
          if (ASMP && v >= res) voff = 0x7ffffff0;
          const int soff = nhwc ? 0 : c * chan_bytes;
          if constexpr (sizeof(OutT) == 4)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), orsrc, voff, soff, 0);
          else
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, o), orsrc, voff, soff, 0);
        }
      }
    }
  }
  };  // run
  constexpr bool kAsm = MTR_WARP_ASM && MTR_WARP_LEAN && !MTR_WARP_ABLATE;
  auto run_planar = [&]() {  // (f32 levels, planar uint8 level 0: one request path)
    if constexpr (kAsm) {
      if (same_shift) { run(std::false_type{}, std::true_type{}); return; }
    }
    run(std::false_type{}, std::false_type{});
  };
  if constexpr (HWC) {
    if (bytes0) run(std::true_type{}, std::false_type{});
    else run_planar();
  } else {
    run_planar();
  }
  }  // items
}

#ifndef MTR_WARP_PERSIST
#define MTR_WARP_PERSIST 0  // developer builds: workgroups per CU of the persistent launch of warp_rows_kernel (0 = one workgroup per tile)
#endif
#ifndef MTR_WARP_ROWS
#define MTR_WARP_ROWS 4  // rows per wave of warp_rows_kernel; 0 = warp_crops_kernel everywhere
#endif

template <typename OutT, int AA, int L0>
static int launch_warp(const void* l0, const float* l1, const float* l2, const float* lut,
                       const LevelDims& dims, unsigned u8_bytes, const float* wp, int n_crops,
                       int res, int nhwc, void* out, hipStream_t stream) {
  constexpr int ROWS = MTR_WARP_ROWS;
  // (warp_crops_kernel keeps the degenerate pyramid levels -- fewer than two rows or columns --
  //  and antialias 4, whose 16 samples per pixel it walks with fewer registers)
  bool degenerate = false;
  for (int l = 0; l < 3; ++l) degenerate |= dims.W[l] < 2 || dims.H[l] < 2;
  if constexpr (ROWS > 0 && AA <= 2) if (!degenerate) {
    constexpr int LX = MTR_WARP_LX, RI = 64 / LX;
    const long long tiles = (long long)((res + LX - 1) / LX) * ((res + MTR_WARP_WAVES * ROWS * RI - 1) / (MTR_WARP_WAVES * ROWS * RI));
    const long long nblocks = (long long)((n_crops + 7) / 8) * 8 * tiles;
    if (nblocks > 0x7fffffffLL) return MTR_E_SHAPE;
    MTR_CLEAR_STALE();
    if constexpr (MTR_WARP_PERSIST > 0) {
      const long long want = 256LL * MTR_WARP_PERSIST;   // (a multiple of 8: whole slots per XCD)
      hipLaunchKernelGGL((warp_rows_kernel<OutT, AA, ROWS ? ROWS : 1, L0, true>), dim3((unsigned)(nblocks < want ? nblocks : want)),
                         dim3(64 * MTR_WARP_WAVES), 0, stream, l0, l1, l2, lut, dims, u8_bytes, wp, n_crops, res, nhwc, (OutT*)out);
    } else {
      hipLaunchKernelGGL((warp_rows_kernel<OutT, AA, ROWS ? ROWS : 1, L0>), dim3((unsigned)nblocks),
                         dim3(64 * MTR_WARP_WAVES), 0, stream, l0, l1, l2, lut, dims, u8_bytes, wp, n_crops, res, nhwc,
                         (OutT*)out);
    }
    MTR_CHECK_LAUNCH();
    return MTR_OK;
  }
  constexpr int PX = MTR_WARP_PX;
  const long long per_crop = (long long)((res + 16 * PX - 1) / (16 * PX)) * ((res + 15) / 16);
  const long long blocks = (long long)((n_crops + 7) / 8) * 8 * per_crop;
  if (blocks > 0x7fffffffLL) return MTR_E_SHAPE;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL((warp_crops_kernel<OutT, AA, PX, L0>), dim3((unsigned)blocks), dim3(256), 0,
                     stream, l0, l1, l2, lut, dims, u8_bytes, wp, n_crops, res, nhwc, (OutT*)out);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

template <typename OutT, int L0>
static int dispatch_warp_aa(const void* l0, const float* l1, const float* l2, const float* lut,
                            const LevelDims& dims, unsigned u8_bytes, const float* wp, int n_crops,
                            int res, int aa, int nhwc, void* out, hipStream_t stream) {
  switch (aa) {
    case 1: return launch_warp<OutT, 1, L0>(l0, l1, l2, lut, dims, u8_bytes, wp, n_crops, res, nhwc, out, stream);
    case 2: return launch_warp<OutT, 2, L0>(l0, l1, l2, lut, dims, u8_bytes, wp, n_crops, res, nhwc, out, stream);
    case 4: return launch_warp<OutT, 4, L0>(l0, l1, l2, lut, dims, u8_bytes, wp, n_crops, res, nhwc, out, stream);
    default: return MTR_E_SHAPE;  // the reference needs torchvision for aa > 4 (:312-315)
  }
}

template <int L0>
static int warp_entry(const void* level0, const float* lut, const float* level1, const float* level2,
                      int N, int Hi, int Wi, const float* warp_params, int n_crops, int res,
                      int antialias, int out_dtype, int out_layout, void* out, hipStream_t s) {
  constexpr bool L0U8 = L0 != 0;
  if (N <= 0 || Hi <= 0 || Wi <= 0 || n_crops < 0 || res <= 0) return MTR_E_SHAPE;
  // a pyramid level of a tiny image may be empty: its pointer may then be NULL
  const bool l1_empty = (Hi / 2) * (Wi / 2) == 0, l2_empty = (Hi / 4) * (Wi / 4) == 0;
  if (!level0 || (!level1 && !l1_empty) || (!level2 && !l2_empty) || !warp_params || !out ||
      (L0U8 && !lut))
    return MTR_E_NULL;
  // a uint8 level 0 is ONE buffer descriptor over the whole frame tensor with 32-bit byte offsets:
  // < 2 GiB per call (the host side, Pose3dEstimator._predict_in_batches, builds the pyramid of the
  // frames one internal batch references when a call's frames exceed that)
  if (L0U8 && (long long)N * 3 * Hi * Wi >= 0x7fffffffLL) return MTR_E_SHAPE;
  // an f32 level 0: the image base is a 64-bit pointer, one image's three planes are addressed with
  // 32-bit byte offsets
  if (!L0U8 && (long long)Hi * Wi * 12 >= 0x7fffffffLL) return MTR_E_SHAPE;
  if (Wi >= (1 << 24) || Hi >= (1 << 24)) return MTR_E_SHAPE;  // (row offsets by v_mul_u32_u24)
  // Range of the uint8 descriptor: the tensor's bytes rounded up to a whole dword (a dword
  // straddling num_records reads as zero, and the byte pair of the last texels may sit in the
  // dword that holds the tensor's last byte).  The <= 3 bytes past the tensor are in the SAME
  // aligned 4-byte word as its last byte -- same page, same allocation granule, whoever allocated
  // it -- and only ever meet zero tap weights; the second dword of a pair beyond that is out of
  // range and returns zero without touching memory.
  const unsigned u8_bytes = (unsigned)(((long long)N * 3 * Hi * Wi + 3) & ~3LL);
  if (out_layout != MTR_NCHW && out_layout != MTR_NHWC) return MTR_E_DTYPE;
  if (n_crops == 0) return MTR_OK;
  if ((uintptr_t)out % 16) return MTR_E_ALIGN;
  if (L0U8 && ((uintptr_t)level0 % 4)) return MTR_E_ALIGN;
  LevelDims dims;
  dims.H[0] = Hi; dims.W[0] = Wi;
  dims.H[1] = Hi / 2; dims.W[1] = Wi / 2;
  dims.H[2] = dims.H[1] / 2; dims.W[2] = dims.W[1] / 2;
  const int nhwc = out_layout == MTR_NHWC;
  switch (out_dtype) {
    case MTR_F32:
      return dispatch_warp_aa<float, L0>(level0, level1, level2, lut, dims, u8_bytes, warp_params,
                                           n_crops, res, antialias, nhwc, out, s);
    case MTR_F16:
      return dispatch_warp_aa<__half, L0>(level0, level1, level2, lut, dims, u8_bytes, warp_params,
                                            n_crops, res, antialias, nhwc, out, s);
    case MTR_BF16:
      return dispatch_warp_aa<__hip_bfloat16, L0>(level0, level1, level2, lut, dims, u8_bytes,
                                                    warp_params, n_crops, res, antialias, nhwc, out, s);
    default: return MTR_E_DTYPE;
  }
}

// ------------------------------------------------------------------------------------------------
// antialias_factor > 4 (multiperson_model.py:312-315): the reference samples the crop at
// res*aa x res*aa and shrinks it with torchvision's antialiased bilinear resize = aten's separable
// _upsample_bilinear2d_aa: a horizontal pass into an f32 intermediate [rows_in, res], then a
// vertical pass; triangle filter of support aa, weights in float with aten's double-typed
// intermediate roundings, normalised by their in-order float sum, taps accumulated as
// t = fma(v_j, w_j, t) from t = 0 (resize_aa.h; bit-exact against torch for K9's resize).  The
// per-crop gamma (crops **= gamma / 2.2, :319) rides on the vertical pass.
// One thread per output value; the 2*aa + 1 taps of neighbouring threads overlap in L1 / L2.
__device__ __forceinline__ float aa_filtered(const float* __restrict__ src, int stride, const AxisSpan& sp) {
  float total = 0.0f;
  for (int j = 0; j < sp.isize; ++j) total = __fadd_rn(total, axis_raw_weight(sp, j, 1));
  float acc = 0.0f;
  for (int j = 0; j < sp.isize; ++j) {
    float w = axis_raw_weight(sp, j, 1);
    if (total != 0.0f) w = __fdiv_rn(w, total);
    acc = __fmaf_rn(src[(size_t)(sp.imin + j) * stride], w, acc);
  }
  return acc;
}

__global__ __launch_bounds__(256) void aa_shrink_rows_kernel(const float* __restrict__ src, long long planes,
                                                             int R, int res, float* __restrict__ tmp) {
  // src [planes, R, R] -> tmp [planes, R, res]
  const long long total = planes * R * res;
  const AxisGeom gx{R, res, 1};
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(t % res);
    const long long row = t / res;  // plane * R + y
    tmp[t] = aa_filtered(src + row * R, 1, axis_span(ox, gx));
  }
}

template <typename OutT>
__global__ __launch_bounds__(256) void aa_shrink_cols_kernel(const float* __restrict__ tmp, int n_crops,
                                                             int R, int res, const float* __restrict__ wp_all,
                                                             int nhwc, OutT* __restrict__ out) {
  // tmp [n_crops*3, R, res] -> out [n_crops, 3, res, res] (or NHWC), ** gamma/2.2 of the crop
  const long long total = (long long)n_crops * 3 * res * res;
  const AxisGeom gy{R, res, 1};
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(t % res), oy = (int)((t / res) % res);
    const long long plane = t / ((long long)res * res);
    const int crop = (int)(plane / 3), c = (int)(plane % 3);
    float v = aa_filtered(tmp + plane * R * res + ox, res, axis_span(oy, gy));
    const float gexp = wp_all[(size_t)crop * MTR_WARP_PARAM_FLOATS + 33];
    if (gexp != 1.0f) v = fast_pow_unit(v, gexp);
    const size_t o = nhwc ? (((size_t)crop * res + oy) * res + ox) * 3 + c : (size_t)t;
    out[o] = from_f32<OutT>(v);
  }
}

}  // namespace mtr

// wide path: whole 8x8 tiles, rows aligned for the 8 / 16 / 8-byte vectors of the three levels
static bool pyramid_wide_ok(const void* src, const void* l1, const void* l2, int Hi, int Wi) {
  return Wi % 8 == 0 && Hi % 8 == 0 && ((uintptr_t)src % 8) == 0 && ((uintptr_t)l1 % 16) == 0 &&
         ((uintptr_t)l2 % 8) == 0;
}
static int pyramid_wide_grid(long long planes, int Hi, int Wi) {
  const long long tiles = planes * (Hi / 8) * (Wi / 8);
  long long grid = (tiles + 255) / 256;
#ifdef MTR_PYR_PER_CU
  constexpr int per_cu = MTR_PYR_PER_CU;  // (developer builds)
#else
  constexpr int per_cu = 160 / MTR_PYR_LUT_COPIES > 8 ? 8 : 160 / MTR_PYR_LUT_COPIES;
#endif
  if (grid > 256 * per_cu) grid = 256 * per_cu;  // persistent: as many workgroups per CU as their LUT copies leave room for (<= 8)
  return (int)grid;
}

static int pyramid_grid(int N, int Hi, int Wi) {
  const long long blocks4 = (long long)N * 3 * ((Hi + 3) / 4) * ((Wi + 3) / 4);
  long long grid = (blocks4 + 255) / 256;
  if (grid > 8192) grid = 8192;  // grid-stride the rest
  return (int)grid;
}

extern "C" int mtr_build_pyramid(const uint8_t* images_u8, int N, int Hi, int Wi, float* level0,
                                 float* level1, float* level2, mtr_stream_t stream) {
  if (N < 0 || Hi <= 0 || Wi <= 0) return MTR_E_SHAPE;
  if (!images_u8 || !level0 || (!level1 && (Hi / 2) * (Wi / 2) > 0) || (!level2 && (Hi / 4) * (Wi / 4) > 0))
    return MTR_E_NULL;
  if (N == 0) return MTR_OK;
  if (((uintptr_t)level0 % 16) || ((uintptr_t)images_u8 % 4)) return MTR_E_ALIGN;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL((mtr::build_pyramid_kernel<true, true>), dim3(pyramid_grid(N, Hi, Wi)), dim3(256),
                     0, (hipStream_t)stream, (const void*)images_u8, N * 3, Hi, Wi, level0, level1,
                     level2, (float*)nullptr, mtr::gamma_lut_host(), mtr::make_fastdiv((unsigned)((Wi + 3) / 4)),
                     mtr::make_fastdiv((unsigned)((Hi + 3) / 4)));
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

template <bool HWC>
static int build_pyramid_u8_entry(const uint8_t* images_u8, int N, int Hi, int Wi, float* lut, float* level1,
                                  float* level2, hipStream_t stream) {
  if (N < 0 || Hi <= 0 || Wi <= 0) return MTR_E_SHAPE;
  if (!images_u8 || !lut || (!level1 && (Hi / 2) * (Wi / 2) > 0) || (!level2 && (Hi / 4) * (Wi / 4) > 0))
    return MTR_E_NULL;
  if (N == 0) return MTR_OK;
  if ((uintptr_t)images_u8 % 4) return MTR_E_ALIGN;
  MTR_CLEAR_STALE();
  if (pyramid_wide_ok(images_u8, level1, level2, Hi, Wi)) {
    // (interleaved frames: a third of the threads, three channels each)
    hipLaunchKernelGGL(mtr::build_pyramid_u8_wide_kernel<HWC>, dim3(pyramid_wide_grid(HWC ? N : 3LL * N, Hi, Wi)),
                       dim3(256), 0, stream, images_u8, HWC ? N : N * 3, Hi, Wi, level1, level2, lut,
                       mtr::gamma_lut_host(), mtr::make_fastdiv((unsigned)(Wi / 8)),
                       mtr::make_fastdiv((unsigned)(Hi / 8)));
    MTR_CHECK_LAUNCH();
    return MTR_OK;
  }
  hipLaunchKernelGGL((mtr::build_pyramid_kernel<true, false, HWC>), dim3(pyramid_grid(N, Hi, Wi)),
                     dim3(256), 0, stream, (const void*)images_u8, N * 3, Hi, Wi,
                     (float*)nullptr, level1, level2, lut, mtr::gamma_lut_host(), mtr::make_fastdiv((unsigned)((Wi + 3) / 4)),
                     mtr::make_fastdiv((unsigned)((Hi + 3) / 4)));
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

extern "C" int mtr_build_pyramid_u8(const uint8_t* images_u8, int N, int Hi, int Wi, float* lut,
                                    float* level1, float* level2, mtr_stream_t stream) {
  return build_pyramid_u8_entry<false>(images_u8, N, Hi, Wi, lut, level1, level2, (hipStream_t)stream);
}

extern "C" int mtr_build_pyramid_u8_hwc(const uint8_t* images_u8, int N, int Hi, int Wi, float* lut,
                                        float* level1, float* level2, mtr_stream_t stream) {
  return build_pyramid_u8_entry<true>(images_u8, N, Hi, Wi, lut, level1, level2, (hipStream_t)stream);
}

extern "C" int mtr_pyramid_from_level0(const float* level0, int N, int Hi, int Wi, float* level1,
                                       float* level2, mtr_stream_t stream) {
  if (N < 0 || Hi <= 0 || Wi <= 0) return MTR_E_SHAPE;
  if (!level0 || (!level1 && (Hi / 2) * (Wi / 2) > 0) || (!level2 && (Hi / 4) * (Wi / 4) > 0))
    return MTR_E_NULL;
  if (N == 0) return MTR_OK;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL((mtr::build_pyramid_kernel<false, false>), dim3(pyramid_grid(N, Hi, Wi)),
                     dim3(256), 0, (hipStream_t)stream, (const void*)level0, N * 3, Hi, Wi,
                     (float*)nullptr, level1, level2, (float*)nullptr, mtr::gamma_lut_host(),
                     mtr::make_fastdiv((unsigned)((Wi + 3) / 4)), mtr::make_fastdiv((unsigned)((Hi + 3) / 4)));
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

extern "C" int mtr_crop_geometry(const float* boxes, int box_stride, const float* intrinsics,
                                 const float* distortion, const float* camspace_up,
                                 const int32_t* image_ids, const float* aug_rotflipmat,
                                 const float* aug_scales, const float* aug_gammas, int n_box,
                                 int n_aug, int res, int antialias, float* new_intrinsics,
                                 float* rot, float* warp_params, mtr_stream_t stream) {
  if (!boxes || !intrinsics || !distortion || !camspace_up || !image_ids || !aug_rotflipmat ||
      !aug_scales || !aug_gammas || !new_intrinsics || !rot || !warp_params)
    return MTR_E_NULL;
  if (n_box < 0 || n_aug <= 0 || res <= 0 || antialias <= 0 || box_stride < 4) return MTR_E_SHAPE;
  if (n_box == 0) return MTR_OK;
  const int total = n_box * n_aug;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(mtr::crop_geometry_kernel, dim3((total + 63) / 64), dim3(64), 0,
                     (hipStream_t)stream, boxes, box_stride, intrinsics, distortion, camspace_up,
                     image_ids, aug_rotflipmat, aug_scales, aug_gammas, n_box, n_aug, res,
                     antialias, new_intrinsics, rot, warp_params);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

extern "C" int mtr_warp_crops(const float* level0, const float* level1, const float* level2, int N,
                              int Hi, int Wi, const float* warp_params, int n_crops, int res,
                              int antialias, int out_dtype, int out_layout, void* out,
                              mtr_stream_t stream) {
  return mtr::warp_entry<0>(level0, nullptr, level1, level2, N, Hi, Wi, warp_params, n_crops, res,
                                antialias, out_dtype, out_layout, out, (hipStream_t)stream);
}

extern "C" int mtr_warp_crops_u8(const uint8_t* level0_u8, const float* lut, const float* level1,
                                 const float* level2, int N, int Hi, int Wi, const float* warp_params,
                                 int n_crops, int res, int antialias, int out_dtype, int out_layout,
                                 void* out, mtr_stream_t stream) {
  return mtr::warp_entry<1>(level0_u8, lut, level1, level2, N, Hi, Wi, warp_params, n_crops, res,
                            antialias, out_dtype, out_layout, out, (hipStream_t)stream);
}

extern "C" int mtr_warp_crops_u8_hwc(const uint8_t* level0_u8, const float* lut, const float* level1,
                                     const float* level2, int N, int Hi, int Wi, const float* warp_params,
                                     int n_crops, int res, int antialias, int out_dtype, int out_layout,
                                     void* out, mtr_stream_t stream) {
  return mtr::warp_entry<2>(level0_u8, lut, level1, level2, N, Hi, Wi, warp_params, n_crops, res,
                            antialias, out_dtype, out_layout, out, (hipStream_t)stream);
}

extern "C" size_t mtr_crops_shrink_workspace_bytes(int n_crops, int res, int antialias) {
  if (n_crops <= 0 || res <= 0 || antialias <= 0) return 0;
  return (size_t)n_crops * 3 * (size_t)(res * antialias) * res * sizeof(float);
}

extern "C" int mtr_crops_shrink_antialiased(const float* crops_big, const float* warp_params, int n_crops,
                                            int res, int antialias, int out_dtype, int out_layout,
                                            void* out, void* workspace, size_t workspace_bytes,
                                            mtr_stream_t stream) {
  if (!crops_big || !warp_params || !out || !workspace) return MTR_E_NULL;
  if (n_crops < 0 || res <= 0 || antialias < 1) return MTR_E_SHAPE;
  if (2 * antialias + 2 > mtr::kDTaps) return MTR_E_SHAPE;  // taps per output value
  if (out_layout != MTR_NCHW && out_layout != MTR_NHWC) return MTR_E_DTYPE;
  if (workspace_bytes < mtr_crops_shrink_workspace_bytes(n_crops, res, antialias)) return MTR_E_WORKSPACE;
  if (n_crops == 0) return MTR_OK;
  const int R = res * antialias;
  hipStream_t s = (hipStream_t)stream;
  const long long planes = (long long)n_crops * 3;
  auto blocks_for = [](long long n) { return (unsigned)((n + 255) / 256 > 65535LL * 16 ? 65535 * 16 : (n + 255) / 256); };
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(mtr::aa_shrink_rows_kernel, dim3(blocks_for(planes * R * res)), dim3(256), 0, s,
                     crops_big, planes, R, res, (float*)workspace);
  MTR_CHECK_LAUNCH();
  const unsigned nb = blocks_for(planes * res * res);
  const int nhwc = out_layout == MTR_NHWC;
  switch (out_dtype) {
    case MTR_F32:
      hipLaunchKernelGGL(mtr::aa_shrink_cols_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)workspace,
                         n_crops, R, res, warp_params, nhwc, (float*)out);
      break;
    case MTR_F16:
      hipLaunchKernelGGL(mtr::aa_shrink_cols_kernel<__half>, dim3(nb), dim3(256), 0, s, (const float*)workspace,
                         n_crops, R, res, warp_params, nhwc, (__half*)out);
      break;
    case MTR_BF16:
      hipLaunchKernelGGL(mtr::aa_shrink_cols_kernel<__hip_bfloat16>, dim3(nb), dim3(256), 0, s,
                         (const float*)workspace, n_crops, R, res, warp_params, nhwc, (__hip_bfloat16*)out);
      break;
    default: return MTR_E_DTYPE;
  }
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}
