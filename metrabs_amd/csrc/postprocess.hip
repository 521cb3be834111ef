// K7 (first "next" row): everything that follows the crop model, in ONE launch.
//
// Replaces the ~15 small launches of
//   Pose3dEstimator._predict_single_batch post-ops  metrabs_pytorch/multiperson/multiperson_model.py:244-259
//     (mirror un-swap through joint_info.mirror_mapping for flipped augs, poses @ R, transpose)
//   Pose3dEstimator._estimate_poses_batched post-ops  multiperson_model.py:143-178
//     (joint_transform_matrix einsum 'bank,nN->baNk', 2D projection K . distort(project(p)),
//      world transform with inv(extrinsics), skeleton index select, mean over the TTA axis)
//
// One workgroup per box; the un-augmented poses of all augs live in LDS ([A][J][3]); arithmetic in
// f64, one rounding to f32 per output (the reference rounds after every f32 op; its own noise at
// 3 m is ~2.4e-4 mm).  Latency-bound, KB-sized.
#include "common.h"

namespace mtr {

struct PostArgs {
  int A, n, J, Jt, S, average;
  int has_transform, has_skeleton;
};

__global__ __launch_bounds__(256) void postprocess_kernel(
    const float* __restrict__ poses_crop, const float* __restrict__ rot,
    const uint8_t* __restrict__ should_flip, const int32_t* __restrict__ mirror,
    const float* __restrict__ jtm, const int32_t* __restrict__ skeleton,
    const float* __restrict__ intr, const float* __restrict__ dist,
    const float* __restrict__ inv_ext, PostArgs a, float* __restrict__ poses3d,
    float* __restrict__ poses2d) {
  extern __shared__ __attribute__((aligned(16))) double q[];  // [A][J][3]
  const int b = blockIdx.x;
  // ---- 1. mirror un-swap (before the back rotation, :249-256) and poses @ R
  for (int t = threadIdx.x; t < a.A * a.J; t += blockDim.x) {
    const int ai = t / a.J, j = t - ai * a.J;
    const int src = should_flip[ai] ? mirror[j] : j;
    const float* p = poses_crop + (((size_t)ai * a.n + b) * a.J + src) * 3;
    const float* R = rot + ((size_t)ai * a.n + b) * 9;
    const double x = p[0], y = p[1], z = p[2];
    q[t * 3 + 0] = x * R[0] + y * R[3] + z * R[6];  // row vector times R
    q[t * 3 + 1] = x * R[1] + y * R[4] + z * R[7];
    q[t * 3 + 2] = x * R[2] + y * R[5] + z * R[8];
  }
  __syncthreads();

  // per-box camera
  double K[6], E[12], d[12];
  bool has_dist = false;
#pragma unroll
  for (int k = 0; k < 6; ++k) K[k] = intr[(size_t)b * 9 + k];
#pragma unroll
  for (int k = 0; k < 12; ++k) E[k] = inv_ext[(size_t)b * 16 + k];
#pragma unroll
  for (int k = 0; k < 12; ++k) { d[k] = dist[(size_t)b * 12 + k]; has_dist |= d[k] != 0.0; }

  const int n_items = a.average ? a.S : a.A * a.S;
  for (int t = threadIdx.x; t < n_items; t += blockDim.x) {
    const int s = a.average ? t : t % a.S;
    const int a0 = a.average ? 0 : t / a.S, a1 = a.average ? a.A : a0 + 1;
    const int jt = a.has_skeleton ? skeleton[s] : s;
    double s3[3] = {0, 0, 0}, s2[2] = {0, 0};
    for (int ai = a0; ai < a1; ++ai) {
      const double* qa = q + (size_t)ai * a.J * 3;
      double P[3];
      if (a.has_transform) {  // 'bank,nN->baNk' (:143-145)
        P[0] = P[1] = P[2] = 0.0;
        for (int j = 0; j < a.J; ++j) {
          const double w = jtm[(size_t)j * a.Jt + jt];
          P[0] += w * qa[j * 3]; P[1] += w * qa[j * 3 + 1]; P[2] += w * qa[j * 3 + 2];
        }
      } else {
        P[0] = qa[jt * 3]; P[1] = qa[jt * 3 + 1]; P[2] = qa[jt * 3 + 2];
      }
      // the reference holds the poses in f32 between the stages
      const double Px = (double)(float)P[0], Py = (double)(float)P[1], Pz = (double)(float)P[2];
      // 2D: K[:2] . [distort(project(P)), 1]  (:148-151; warping.distort_points warping.py:57-62)
      double x = Px / Pz, y = Py / Pz;
      if (has_dist) {
        const double r2 = x * x + y * y;
        const double ra = (((d[4] * r2 + d[1]) * r2 + d[0]) * r2 + 1.0) /
                          (((d[7] * r2 + d[6]) * r2 + d[5]) * r2 + 1.0);
        const double rb = 2.0 * (x * d[3] + y * d[2]);
        const double cx = (d[9] * r2 + d[3] + d[8]) * r2, cy = (d[11] * r2 + d[2] + d[10]) * r2;
        const double nx = x * (ra + rb) + cx, ny = y * (ra + rb) + cy;
        x = nx; y = ny;
      }
      const double u = K[0] * x + K[1] * y + K[2], v = K[3] * x + K[4] * y + K[5];
      // 3D: inv(extrinsics)[:3] . [P, 1]  (:167-170)
      const double wx = E[0] * Px + E[1] * Py + E[2] * Pz + E[3];
      const double wy = E[4] * Px + E[5] * Py + E[6] * Pz + E[7];
      const double wz = E[8] * Px + E[9] * Py + E[10] * Pz + E[11];
      // per-aug values are f32 tensors in the reference; the TTA mean averages those
      s3[0] += (double)(float)wx; s3[1] += (double)(float)wy; s3[2] += (double)(float)wz;
      s2[0] += (double)(float)u; s2[1] += (double)(float)v;
    }
    const double inv = a.average ? 1.0 / a.A : 1.0;
    const size_t o = (size_t)b * n_items + t;
    poses3d[o * 3 + 0] = (float)(s3[0] * inv);
    poses3d[o * 3 + 1] = (float)(s3[1] * inv);
    poses3d[o * 3 + 2] = (float)(s3[2] * inv);
    poses2d[o * 2 + 0] = (float)(s2[0] * inv);
    poses2d[o * 2 + 1] = (float)(s2[1] * inv);
  }
}

// Metrabs.latent_points_to_joints (metrabs_tf/models/metrabs.py:80-81 -> tfu3d.linear_combine_points
// tfu3d.py:48-49, einsum 'bjc,jJ->bJc'; called by Metrabs.forward behind reconstruct_absolute when
// transform_coords / predict_all_and_latents, metrabs_pytorch/models/metrabs.py:61-62): every output
// joint is an affine combination of the crop's latent points.  One workgroup per crop, the crop's
// points in LDS, one thread per (output joint, coordinate); the weight column of a joint is read
// with lanes along J_out (coalesced), sums in f64, one rounding.
__global__ __launch_bounds__(256) void linear_combine_kernel(
    const float* __restrict__ points, const float* __restrict__ weights, int Jin, int Jout,
    float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float pts[];  // [Jin][3]
  const int b = blockIdx.x;
  for (int t = threadIdx.x; t < Jin * 3; t += blockDim.x) pts[t] = points[(size_t)b * Jin * 3 + t];
  __syncthreads();
  for (int jo = threadIdx.x; jo < Jout; jo += blockDim.x) {
    double x = 0.0, y = 0.0, z = 0.0;
    for (int j = 0; j < Jin; ++j) {
      const double w = weights[(size_t)j * Jout + jo];
      x += w * pts[j * 3]; y += w * pts[j * 3 + 1]; z += w * pts[j * 3 + 2];
    }
    float* o = out + ((size_t)b * Jout + jo) * 3;
    o[0] = (float)x; o[1] = (float)y; o[2] = (float)z;
  }
}

}  // namespace mtr

extern "C" int mtr_linear_combine_points(const float* points, const float* weights, int B, int J_in,
                                         int J_out, float* out, mtr_stream_t stream) {
  if (!points || !weights || !out) return MTR_E_NULL;
  if (B < 0 || J_in <= 0 || J_out <= 0) return MTR_E_SHAPE;
  const size_t lds = (size_t)J_in * 3 * sizeof(float);
  if (lds > 48 * 1024) return MTR_E_SHAPE;  // <= 4096 latent points
  if (B == 0) return MTR_OK;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(mtr::linear_combine_kernel, dim3(B), dim3(J_out > 128 ? 256 : (J_out > 64 ? 128 : 64)),
                     lds, (hipStream_t)stream, points, weights, J_in, J_out, out);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

extern "C" int mtr_postprocess_poses(const float* poses_crop, const float* rot,
                                     const uint8_t* should_flip, const int32_t* mirror_mapping,
                                     const float* joint_transform, int Jt, const int32_t* skeleton,
                                     int S, const float* intrinsics, const float* distortion,
                                     const float* inv_extrinsics, int A, int n, int J,
                                     int average_aug, float* poses3d, float* poses2d,
                                     mtr_stream_t stream) {
  if (!poses_crop || !rot || !should_flip || !mirror_mapping || !intrinsics || !distortion ||
      !inv_extrinsics || !poses3d || !poses2d)
    return MTR_E_NULL;
  if (A <= 0 || n < 0 || J <= 0) return MTR_E_SHAPE;
  if (joint_transform && Jt <= 0) return MTR_E_SHAPE;
  if (skeleton && S <= 0) return MTR_E_SHAPE;
  if (n == 0) return MTR_OK;
  mtr::PostArgs a;
  a.A = A; a.n = n; a.J = J;
  a.has_transform = joint_transform != nullptr;
  a.has_skeleton = skeleton != nullptr;
  a.Jt = a.has_transform ? Jt : J;
  a.S = a.has_skeleton ? S : a.Jt;
  a.average = average_aug != 0;
  const size_t lds = (size_t)A * J * 3 * sizeof(double);
  if (lds > 144 * 1024) return MTR_E_SHAPE;  // A*J <= 6144 (one box's poses live in LDS)
  if (lds > 48 * 1024) {  // e.g. the 555-point multi-skeleton heads with num_aug = 5
    const int rc = mtr::allow_dynamic_lds((const void*)mtr::postprocess_kernel, lds);
    if (rc != MTR_OK) return rc;
  }
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(mtr::postprocess_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream,
                     poses_crop, rot, should_flip, mirror_mapping, joint_transform, skeleton,
                     intrinsics, distortion, inv_extrinsics, a, poses3d, poses2d);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}
