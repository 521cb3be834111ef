// K5: absolute (camera-space) pose reconstruction.
//
// Replaces ptu3d.reconstruct_absolute (metrabs_pytorch/ptu3d.py:9-33) with
// reconstruct_ref_fullpersp (ptu3d.py:56-105), reconstruct_ref_weakpersp (ptu3d.py:36-49),
// is_within_fov (ptu3d.py:113-121) and back_project (ptu3d.py:108-110): ~40 tiny launches and a
// batched LAPACK call in the reference, two launches here.
//
//   0. small batches (B*J <= 16384, i.e. every internal batch of the reference): ONE launch, each
//      workgroup recomputes the batch moments redundantly (recon_fused_small_kernel);
//   1. moments kernel: sum(normalized2d^2) and sum(rel_backproj^2) over the WHOLE call batch --
//      the reference's rms_normalize is batch-global (ptu3d.py:71-74).  <= 256 per-block fp64
//      partials, combined in a fixed order (no atomics: run-to-run deterministic).
//   2. solve kernel: one wave per crop.  The reference stacks 2J weighted rows [1 0 -x; 0 1 -y]
//      plus three ridge rows sqrt(l2)*I and calls torch.linalg.lstsq; the same minimiser is the
//      solution of the 3x3 normal equations  (A^T W^2 A + l2*I) r = A^T W^2 b, which are formed
//      with fp64 accumulators and solved in closed form in fp64, then rounded to fp32.
//
// Latency-bound (KB-sized): report microseconds, not a roofline fraction.
#include "common.h"

namespace mtr {

constexpr int kMaxMomentBlocks = 256;

struct Cam {
  double k00, k01, k02, k10, k11, k12;  // first two rows of K^-1
};

// first two rows of inv(K) for a general 3x3 (adjugate / determinant, fp64)
__device__ __forceinline__ Cam inverse_rows01(const float* __restrict__ K) {
  const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7],
               i = K[8];
  const double A = e * i - f * h, Bc = -(d * i - f * g), C = d * h - e * g;
  const double det = a * A + b * Bc + c * C;
  const double inv = 1.0 / det;
  Cam r;
  r.k00 = A * inv;
  r.k01 = -(b * i - c * h) * inv;
  r.k02 = (b * f - c * e) * inv;
  r.k10 = Bc * inv;
  r.k11 = (a * i - c * g) * inv;
  r.k12 = -(a * f - c * d) * inv;
  return r;
}

// normalized image coordinates: ([x, y, 1] @ K^-T)[:2]  (ptu3d.py:13)
__device__ __forceinline__ void normalize2d(const Cam& c, float x, float y, double& nx, double& ny) {
  nx = c.k00 * x + c.k01 * y + c.k02;
  ny = c.k10 * x + c.k11 * y + c.k12;
}

__global__ __launch_bounds__(256) void recon_moments_kernel(
    const float* __restrict__ coords2d, const float* __restrict__ rel,
    const float* __restrict__ intr, int B, int J, double* __restrict__ partials) {
  __shared__ double red[2][4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double s2d = 0.0, srb = 0.0;
  for (int b = blockIdx.x * 4 + wid; b < B; b += gridDim.x * 4) {
    const Cam cam = inverse_rows01(intr + (size_t)b * 9);
    for (int j = lane; j < J; j += 64) {
      const size_t o = (size_t)b * J + j;
      double nx, ny;
      normalize2d(cam, coords2d[o * 2], coords2d[o * 2 + 1], nx, ny);
      const double rx = rel[o * 3], ry = rel[o * 3 + 1], rz = rel[o * 3 + 2];
      const double bx = nx * rz - rx, by = ny * rz - ry;
      s2d += nx * nx + ny * ny;
      srb += bx * bx + by * by;
    }
  }
  s2d = group_sum<64>(s2d);
  srb = group_sum<64>(srb);
  if (lane == 0) { red[0][wid] = s2d; red[1][wid] = srb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partials[blockIdx.x * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// partials[n][2] -> moments[3] = {sum2d, sumrb, count}; one wave, fixed summation order
__global__ __launch_bounds__(64) void recon_finalize_kernel(const double* __restrict__ partials,
                                                            int n, double count,
                                                            double* __restrict__ moments) {
  const int lane = threadIdx.x;
  double a = 0.0, b = 0.0;
  for (int i = lane; i < n; i += 64) { a += partials[i * 2]; b += partials[i * 2 + 1]; }
  a = group_sum<64>(a);
  b = group_sum<64>(b);
  if (lane == 0) { moments[0] = a; moments[1] = b; moments[2] = count; }
}

__device__ __forceinline__ bool within_fov(float x, float y, float lower, float upper) {
  return x >= lower && x <= upper && y >= lower && y <= upper;
}

// Solve the symmetric 3x3 system M r = v (Cramer, fp64).
__device__ __forceinline__ void solve_sym3(double m00, double m01, double m02, double m11,
                                           double m12, double m22, double v0, double v1, double v2,
                                           double& r0, double& r1, double& r2) {
  const double c00 = m11 * m22 - m12 * m12;
  const double c01 = m02 * m12 - m01 * m22;
  const double c02 = m01 * m12 - m02 * m11;
  const double det = m00 * c00 + m01 * c01 + m02 * c02;
  const double c11 = m00 * m22 - m02 * m02;
  const double c12 = m01 * m02 - m00 * m12;
  const double c22 = m00 * m11 - m01 * m01;
  const double inv = 1.0 / det;
  r0 = (c00 * v0 + c01 * v1 + c02 * v2) * inv;
  r1 = (c01 * v0 + c11 * v1 + c12 * v2) * inv;
  r2 = (c02 * v0 + c12 * v1 + c22 * v2) * inv;
}

struct ReconArgs {
  float fov_lower, fov_upper;
  float mix;
  int mix_enabled;
  int weak;
  float l2_reg, weight_eps;
};

// one wave solves one crop; m0/m1/m2 = sum(normalized2d^2), sum(rel_backproj^2), count
__device__ __forceinline__ void recon_solve_crop(
    const float* __restrict__ coords2d, const float* __restrict__ rel,
    const float* __restrict__ intr, int b, int lane, int J, const ReconArgs& a, double m0, double m1,
    double m2, float* __restrict__ poses) {
  const Cam cam = inverse_rows01(intr + (size_t)b * 9);
  const size_t base = (size_t)b * J;

  double ref0, ref1, ref2;
  if (!a.weak) {
    // rms over the whole call batch (ptu3d.py:71-74, 82, 90)
    const double scale2d = sqrt(m0 / m2);
    const double scale_rb = sqrt(m1 / m2);
    double m02 = 0, m12 = 0, m22 = 0, sw = 0, v0 = 0, v1 = 0, v2 = 0;
    for (int j = lane; j < J; j += 64) {
      const float px = coords2d[(base + j) * 2], py = coords2d[(base + j) * 2 + 1];
      double nx, ny;
      normalize2d(cam, px, py, nx, ny);
      const double rx = rel[(base + j) * 3], ry = rel[(base + j) * 3 + 1],
                   rz = rel[(base + j) * 3 + 2];
      const double xh = nx / scale2d, yh = ny / scale2d;
      const double bx = (nx * rz - rx) / scale_rb, by = (ny * rz - ry) / scale_rb;
      // weights = mask.float() + 1e-4 in fp32 (ptu3d.py:94), squared by the normal equations
      const float wf = (within_fov(px, py, a.fov_lower, a.fov_upper) ? 1.0f : 0.0f) + a.weight_eps;
      const double w2 = (double)wf * (double)wf;
      sw += w2;
      m02 -= w2 * xh;
      m12 -= w2 * yh;
      m22 += w2 * (xh * xh + yh * yh);
      v0 += w2 * bx;
      v1 += w2 * by;
      v2 -= w2 * (xh * bx + yh * by);
    }
    sw = group_sum<64>(sw);
    m02 = group_sum<64>(m02);
    m12 = group_sum<64>(m12);
    m22 = group_sum<64>(m22);
    v0 = group_sum<64>(v0);
    v1 = group_sum<64>(v1);
    v2 = group_sum<64>(v2);
    // ridge rows sqrt(l2)*I with zero rhs (ptu3d.py:79-80,91-92,95-98)
    const double l2 = (double)sqrtf(a.l2_reg) * (double)sqrtf(a.l2_reg);
    double r0, r1, r2;
    solve_sym3(sw + l2, 0.0, m02, sw + l2, m12, m22 + l2, v0, v1, v2, r0, r1, r2);
    ref0 = r0 * scale_rb;
    ref1 = r1 * scale_rb;
    ref2 = r2 * (scale_rb / scale2d);
  } else {
    // weak perspective (ptu3d.py:36-49; ptu.mean_stdev_masked ptu.py:4-19): masked means and
    // a single stdev over both image axes, per crop.
    double n = 0, s3x = 0, s3y = 0, s3z = 0, s2x = 0, s2y = 0;
    for (int j = lane; j < J; j += 64) {
      const float px = coords2d[(base + j) * 2], py = coords2d[(base + j) * 2 + 1];
      if (!within_fov(px, py, a.fov_lower, a.fov_upper)) continue;
      double nx, ny;
      normalize2d(cam, px, py, nx, ny);
      n += 1.0;
      s3x += rel[(base + j) * 3]; s3y += rel[(base + j) * 3 + 1]; s3z += rel[(base + j) * 3 + 2];
      s2x += nx; s2y += ny;
    }
    n = group_sum<64>(n);
    s3x = group_sum<64>(s3x); s3y = group_sum<64>(s3y); s3z = group_sum<64>(s3z);
    s2x = group_sum<64>(s2x); s2y = group_sum<64>(s2y);
    // nan_to_num(x / 0) -> 0 when no joint is valid
    const double m3x = n > 0 ? s3x / n : 0.0, m3y = n > 0 ? s3y / n : 0.0, m3z = n > 0 ? s3z / n : 0.0;
    const double m2x = n > 0 ? s2x / n : 0.0, m2y = n > 0 ? s2y / n : 0.0;
    double ss3 = 0, ss2 = 0;
    for (int j = lane; j < J; j += 64) {
      const float px = coords2d[(base + j) * 2], py = coords2d[(base + j) * 2 + 1];
      if (!within_fov(px, py, a.fov_lower, a.fov_upper)) continue;
      double nx, ny;
      normalize2d(cam, px, py, nx, ny);
      const double dx3 = rel[(base + j) * 3] - m3x, dy3 = rel[(base + j) * 3 + 1] - m3y;
      ss3 += dx3 * dx3 + dy3 * dy3;
      ss2 += (nx - m2x) * (nx - m2x) + (ny - m2y) * (ny - m2y);
    }
    ss3 = group_sum<64>(ss3);
    ss2 = group_sum<64>(ss2);
    double sd3 = sqrt((n > 0 ? ss3 / n : 0.0) + 1e-10);
    double sd2 = sqrt((n > 0 ? ss2 / n : 0.0) + 1e-10);
    sd3 = fmax(sd3, 1e-5);
    sd2 = fmax(sd2, 1e-5);
    const double z = sd3 / sd2;
    ref0 = m2x * z - m3x;
    ref1 = m2y * z - m3y;
    ref2 = z - m3z;
  }

  const float rf0 = (float)ref0, rf1 = (float)ref1, rf2 = (float)ref2;
  for (int j = lane; j < J; j += 64) {
    const float px = coords2d[(base + j) * 2], py = coords2d[(base + j) * 2 + 1];
    double nx, ny;
    normalize2d(cam, px, py, nx, ny);
    const float rx = rel[(base + j) * 3], ry = rel[(base + j) * 3 + 1], rz = rel[(base + j) * 3 + 2];
    // coords_abs_3d_based = coords3d_rel + ref (ptu3d.py:22)
    const double ax = (double)rx + rf0, ay = (double)ry + rf1, az = (double)rz + rf2;
    float ox = (float)ax, oy = (float)ay, oz = (float)az;
    if (within_fov(px, py, a.fov_lower, a.fov_upper)) {
      // back_project (ptu3d.py:108-110): [nx, ny, 1] * (rel_z + ref_z)
      const double depth = (double)(float)az;
      double bx = (double)(float)nx * depth, by = (double)(float)ny * depth, bz = depth;
      if (a.mix_enabled) {
        const double m = a.mix, om = 1.0 - (double)a.mix;
        bx = m * (double)ox + om * bx;
        by = m * (double)oy + om * by;
        bz = m * (double)oz + om * bz;
      }
      ox = (float)bx; oy = (float)by; oz = (float)bz;
    }
    poses[(base + j) * 3 + 0] = ox;
    poses[(base + j) * 3 + 1] = oy;
    poses[(base + j) * 3 + 2] = oz;
  }
}


__global__ __launch_bounds__(256) void recon_solve_kernel(
    const float* __restrict__ coords2d, const float* __restrict__ rel,
    const float* __restrict__ intr, int B, int J, ReconArgs a, const double* __restrict__ moments,
    float* __restrict__ poses) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const double m0 = a.weak ? 0.0 : moments[0], m1 = a.weak ? 0.0 : moments[1],
               m2 = a.weak ? 1.0 : moments[2];
  recon_solve_crop(coords2d, rel, intr, b, lane, J, a, m0, m1, m2, poses);
}

// Small batches (B*J <= kFusedMaxElems): ONE launch.  Every workgroup recomputes the batch moments
// itself (the whole batch's coordinates are a few tens of KB, L2-resident) in a fixed order, so all
// workgroups hold bit-identical scalars without any inter-workgroup hand-off, then solves its 4 crops.
constexpr int kFusedMaxElems = 16384;

__global__ __launch_bounds__(256) void recon_fused_small_kernel(
    const float* __restrict__ coords2d, const float* __restrict__ rel,
    const float* __restrict__ intr, int B, int J, ReconArgs a, float* __restrict__ poses) {
  __shared__ double red[2][4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double m0 = 0.0, m1 = 0.0;
  if (!a.weak) {
    // (crop, joint) pairs flattened over the 256 threads: a wave per crop left 47 of 64 lanes idle
    // at J = 17 and walked B / 4 crops one after the other (9.5 us at B = 64; the camera inverse is
    // recomputed per pair -- cheaper than a pass through LDS and a barrier)
    double s2d = 0.0, srb = 0.0;
    for (int o = threadIdx.x; o < B * J; o += 256) {
      const Cam cam = inverse_rows01(intr + (size_t)(o / J) * 9);
      double nx, ny;
      normalize2d(cam, coords2d[(size_t)o * 2], coords2d[(size_t)o * 2 + 1], nx, ny);
      const double rx = rel[(size_t)o * 3], ry = rel[(size_t)o * 3 + 1], rz = rel[(size_t)o * 3 + 2];
      const double bx = nx * rz - rx, by = ny * rz - ry;
      s2d += nx * nx + ny * ny;
      srb += bx * bx + by * by;
    }
    s2d = group_sum<64>(s2d);
    srb = group_sum<64>(srb);
    if (lane == 0) { red[0][wid] = s2d; red[1][wid] = srb; }
    __syncthreads();
    m0 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    m1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
  const int b = blockIdx.x * 4 + wid;
  if (b >= B) return;
  recon_solve_crop(coords2d, rel, intr, b, lane, J, a, m0, m1, (double)B * J * 2.0, poses);
}

static ReconArgs make_args(const mtr_recon_params& p) {
  ReconArgs a;
  // is_within_fov (ptu3d.py:113-121)
  const float offset = p.centered_stride ? 0.0f : -(float)p.stride_train / 2.0f;
  a.fov_lower = (float)p.stride_train * p.fov_border_factor + offset;
  a.fov_upper = (float)p.proc_side - (float)p.stride_train * p.fov_border_factor + offset;
  a.mix = p.mix_3d_inside_fov;
  a.mix_enabled = p.mix_enabled;
  a.weak = p.weak_perspective;
  a.l2_reg = p.l2_reg;
  a.weight_eps = p.weight_eps;
  return a;
}

static int moment_blocks(int B) {
  const int need = (B + 3) / 4;
  return need < kMaxMomentBlocks ? (need < 1 ? 1 : need) : kMaxMomentBlocks;
}

}  // namespace mtr

extern "C" size_t mtr_reconstruct_workspace_bytes(int B, int J) {
  (void)J;
  if (B <= 0) return 64;
  // [moments: 3 doubles, padded to 4][partials: blocks x 2 doubles]
  return (size_t)(4 + 2 * mtr::moment_blocks(B)) * sizeof(double);
}

static int check_recon_common(const float* c2d, const float* rel, const float* intr, int B, int J) {
  if (!c2d || !rel || !intr) return MTR_E_NULL;
  if (B < 0 || J <= 0) return MTR_E_SHAPE;
  return MTR_OK;
}

extern "C" int mtr_reconstruct_moments(const float* coords2d, const float* coords3d_rel,
                                       const float* intrinsics, int B, int J, double* moments,
                                       void* workspace, size_t workspace_bytes,
                                       mtr_stream_t stream) {
  int rc = check_recon_common(coords2d, coords3d_rel, intrinsics, B, J);
  if (rc) return rc;
  if (!moments || !workspace) return MTR_E_NULL;
  if (workspace_bytes < mtr_reconstruct_workspace_bytes(B, J) || ((uintptr_t)workspace % 16))
    return MTR_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  double* partials = (double*)workspace + 4;
  const int blocks = mtr::moment_blocks(B);
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(mtr::recon_moments_kernel, dim3(blocks), dim3(256), 0, s, coords2d,
                     coords3d_rel, intrinsics, B, J, partials);
  MTR_CHECK_LAUNCH();
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(mtr::recon_finalize_kernel, dim3(1), dim3(64), 0, s, partials, blocks,
                     (double)B * (double)J * 2.0, moments);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

extern "C" int mtr_reconstruct_solve(const float* coords2d, const float* coords3d_rel,
                                     const float* intrinsics, int B, int J,
                                     const mtr_recon_params* p, const double* moments,
                                     float* poses3d, mtr_stream_t stream) {
  int rc = check_recon_common(coords2d, coords3d_rel, intrinsics, B, J);
  if (rc) return rc;
  if (!p || !poses3d) return MTR_E_NULL;
  if (!p->weak_perspective && !moments) return MTR_E_NULL;
  if (p->proc_side <= 0 || p->stride_train <= 0) return MTR_E_PARAM;
  if (B == 0) return MTR_OK;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(mtr::recon_solve_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     coords2d, coords3d_rel, intrinsics, B, J, mtr::make_args(*p), moments,
                     poses3d);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

extern "C" int mtr_reconstruct_absolute(const float* coords2d, const float* coords3d_rel,
                                        const float* intrinsics, int B, int J,
                                        const mtr_recon_params* p, float* poses3d, void* workspace,
                                        size_t workspace_bytes, mtr_stream_t stream) {
  int rc = check_recon_common(coords2d, coords3d_rel, intrinsics, B, J);
  if (rc) return rc;
  if (!p || !poses3d || !workspace) return MTR_E_NULL;
  if (B == 0) return MTR_OK;
  if (p->proc_side <= 0 || p->stride_train <= 0) return MTR_E_PARAM;
  if ((long long)B * J <= mtr::kFusedMaxElems) {
    MTR_CLEAR_STALE();
    hipLaunchKernelGGL(mtr::recon_fused_small_kernel, dim3((B + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, coords2d, coords3d_rel, intrinsics, B, J,
                       mtr::make_args(*p), poses3d);
    MTR_CHECK_LAUNCH();
    return MTR_OK;
  }
  double* moments = (double*)workspace;
  if (!p->weak_perspective) {
    rc = mtr_reconstruct_moments(coords2d, coords3d_rel, intrinsics, B, J, moments, workspace,
                                 workspace_bytes, stream);
    if (rc) return rc;
  } else if (workspace_bytes < 32) {
    return MTR_E_WORKSPACE;
  }
  return mtr_reconstruct_solve(coords2d, coords3d_rel, intrinsics, B, J, p, moments, poses3d,
                               stream);
}
