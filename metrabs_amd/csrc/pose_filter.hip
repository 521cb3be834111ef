// K8: plausibility filter + pose NMS -- the step behind the hot path
// inside detect_poses(suppress_implausible_poses=True).
//
// Replaces, in one launch per call (one workgroup per image):
//   TF  metrabs_tf/multiperson/multiperson_model.py:441-459   _filter_poses
//   TF  metrabs_tf/multiperson/plausibility_check.py:9-96     / PyTorch port
//       metrabs_pytorch/multiperson/plausibility_check.py:8-119:
//         is_pose_plausible                 bones vs mean bone lengths
//         are_augmentation_results_consistent, scale_align, point_stdev
//         is_pose_consistent_with_box
//         compute_pose_similarity, pose_non_max_suppression, non_max_suppression_overlaps
// (the PyTorch reference keeps the call site commented out, multiperson_model.py:158-163, and its
//  is_pose_consistent_with_box raises as written; the algorithm is the twins' shared one, the two
//  places where they differ are parameters: variance correction and output order).
//
// Data is KB-sized (<= a few hundred poses of <= a few hundred joints): latency-bound by nature;
// the point of the kernel is one launch instead of ~40 small torch ops and a Python loop with a
// device-to-host sync per image (the greedy NMS is sequential in the reference).  All reductions
// are f64 internally, so the float results sit within 1 ulp of an exactly rounded evaluation.
#include "common.h"

namespace mtr {

struct FilterArgs {
  int A, J, J_model, n_bones;
  float rel_small, rel_big, abs_diff_mm;   // 0.1, 3, 300
  float stdev_mm;                          // 200
  float box_fraction;                      // 0.5
  float sim_scale_mm, sim_threshold;       // 300, 0.4
  int max_output, var_correction, order_by_score;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one workgroup (256 threads = 4 waves) per image
__global__ __launch_bounds__(256) void pose_filter_kernel(
    const float* __restrict__ poses3d,  // [P, A, J, 3]
    const float* __restrict__ poses2d,  // [P, A, J, 2]
    const float* __restrict__ boxes,    // [P, 5] x, y, w, h, score
    const int32_t* __restrict__ row_start,  // [n_images + 1]
    const int32_t* __restrict__ edges,      // [n_bones, 2]
    const float* __restrict__ mean_bones,   // [n_bones]
    FilterArgs fa,
    float* __restrict__ ws_mean,  // [P, J, 3] mean-over-aug poses
    float* __restrict__ ws_sim,   // [sum n_i^2] similarity matrices, image i at offset sim_off[i]
    const int32_t* __restrict__ sim_off,  // [n_images]
    uint8_t* __restrict__ valid_out,  // [P]
    int32_t* __restrict__ keep_idx,   // [P]: per image its kept poses (global indices), then -1
    int32_t* __restrict__ keep_count  // [n_images]
) {
  // dynamic LDS: [4 waves][J rounded up to even] float distances, then [4 waves][A] double factors
  extern __shared__ __attribute__((aligned(16))) float dynf[];
  __shared__ int s_valid[1024], s_list[1024], s_rank[1024], s_sel[1024];
  __shared__ float s_scale[1024];
  __shared__ int s_nv, s_nsel;

  const int img = blockIdx.x;
  const int p0 = row_start[img], n = row_start[img + 1] - p0;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int A = fa.A, J = fa.J;
  if (n <= 0) {
    if (tid == 0) keep_count[img] = 0;
    return;
  }
  // ---- 1. per pose (one wave each): mean over augmentations, the three plausibility tests
  for (int q = wid; q < n; q += 4) {
    const int p = p0 + q;
    const float* P3 = poses3d + (size_t)p * A * J * 3;
    const float* P2 = poses2d + (size_t)p * A * J * 2;
    // mean pose (torch.mean over the aug axis: sum in order, then divide)
    for (int e = lane; e < J * 3; e += 64) {
      float s = 0.0f;
      for (int a = 0; a < A; ++a) s = __fadd_rn(s, P3[(size_t)a * J * 3 + e]);
      ws_mean[(size_t)p * J * 3 + e] = __fdiv_rn(s, (float)A);
    }
    __threadfence();  // the other lanes of this wave read the mean pose back below
    const float* M = ws_mean + (size_t)p * J * 3;
    // (a) bones
    int bad = 0;
    for (int b = lane; b < fa.n_bones; b += 64) {
      const int j1 = edges[b * 2], j2 = edges[b * 2 + 1];
      if (j1 >= fa.J_model || j2 >= fa.J_model) continue;
      const double dx = (double)M[j1 * 3] - M[j2 * 3], dy = (double)M[j1 * 3 + 1] - M[j2 * 3 + 1],
                   dz = (double)M[j1 * 3 + 2] - M[j2 * 3 + 2];
      const float len = (float)sqrt(dx * dx + dy * dy + dz * dz);
      const float rel = __fdiv_rn(len, mean_bones[b]);
      const float diff = fabsf(len - mean_bones[b]);
      if ((rel > fa.rel_big || rel < fa.rel_small) && diff > fa.abs_diff_mm) bad = 1;
    }
    bad = __any(bad);
    // (b) augmentation consistency: scale-align the A results, stdev of every joint over A.
    // The per-aug scale factors sqrt(mean_sq / sq_a) go through this wave's slice of LDS.
    int n_stable = 0;
    {
      double* fac = reinterpret_cast<double*>(dynf + 4 * ((J + 1) & ~1)) + wid * A;
      double msq = 0.0;
      for (int a = 0; a < A; ++a) {
        double s = 0.0;
        for (int e = lane; e < J * 3; e += 64) {
          const double v = P3[(size_t)a * J * 3 + e];
          s += v * v;
        }
        s = wave_sum(s) / (double)(J * 3);
        if (lane == 0) fac[a] = s;
        msq += s;
      }
      msq /= (double)A;
      __builtin_amdgcn_wave_barrier();
      for (int a = lane; a < A; a += 64) fac[a] = sqrt(msq / fac[a]);
      __builtin_amdgcn_wave_barrier();
      for (int j = lane; j < J; j += 64) {
        double var_sum = 0.0;
        for (int cdim = 0; cdim < 3; ++cdim) {
          double mean = 0.0;
          for (int a = 0; a < A; ++a) mean += (double)P3[((size_t)a * J + j) * 3 + cdim] * fac[a];
          mean /= (double)A;
          double ss = 0.0;
          for (int a = 0; a < A; ++a) {
            const double d = (double)P3[((size_t)a * J + j) * 3 + cdim] * fac[a] - mean;
            ss += d * d;
          }
          var_sum += ss / (double)(A - fa.var_correction);  // A == 1 with correction 1: NaN, as torch
        }
        const float sd = (float)sqrt(var_sum);
        if (sd < fa.stdev_mm) ++n_stable;
      }
      for (int o = 32; o > 0; o >>= 1) n_stable += __shfl_xor(n_stable, o);
    }
    const bool consistent = n_stable > J / 4;
    // (c) box consistency on the mean 2D pose
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int j = lane; j < J; j += 64) {
      float sx = 0.0f, sy = 0.0f;
      for (int a = 0; a < A; ++a) {
        sx = __fadd_rn(sx, P2[((size_t)a * J + j) * 2]);
        sy = __fadd_rn(sy, P2[((size_t)a * J + j) * 2 + 1]);
      }
      sx = __fdiv_rn(sx, (float)A);
      sy = __fdiv_rn(sy, (float)A);
      mnx = fminf(mnx, sx); mxx = fmaxf(mxx, sx);
      mny = fminf(mny, sy); mxy = fmaxf(mxy, sy);
    }
    for (int o = 32; o > 0; o >>= 1) {
      mnx = fminf(mnx, __shfl_xor(mnx, o)); mxx = fmaxf(mxx, __shfl_xor(mxx, o));
      mny = fminf(mny, __shfl_xor(mny, o)); mxy = fmaxf(mxy, __shfl_xor(mxy, o));
    }
    const float bx = boxes[p * 5], by = boxes[p * 5 + 1], bw = boxes[p * 5 + 2], bh = boxes[p * 5 + 3];
    const float ix = fmaxf(__fsub_rn(fminf(__fadd_rn(bx, bw), mxx), fmaxf(bx, mnx)), 0.0f);
    const float iy = fmaxf(__fsub_rn(fminf(__fadd_rn(by, bh), mxy), fmaxf(by, mny)), 0.0f);
    const bool in_box = __fmul_rn(ix, iy) > __fmul_rn(fa.box_fraction, __fmul_rn(bw, bh));
    // square scale of the mean pose (compute_pose_similarity)
    double s = 0.0;
    for (int e = lane; e < J * 3; e += 64) {
      const double v = M[e];
      s += v * v;
    }
    s = wave_sum(s) / (double)(J * 3);
    if (lane == 0) {
      const int ok = (!bad && consistent && in_box) ? 1 : 0;
      if (q < 1024) {
        s_valid[q] = ok;
        s_scale[q] = (float)s;
      }
      valid_out[p] = (uint8_t)ok;
    }
  }
  __syncthreads();
  // ---- 2. valid poses in index order
  if (tid == 0) {
    int nv = 0;
    for (int q = 0; q < n && q < 1024; ++q)
      if (s_valid[q]) s_list[nv++] = q;
    s_nv = nv;
  }
  __syncthreads();
  const int nv = s_nv;
  // ---- 3. similarity matrix of the valid poses: one wave per ordered pair (u <= v mirrored)
  float* S = ws_sim + sim_off[img];
  float* dist = dynf + wid * ((J + 1) & ~1);
  const int k = J / 4;
  for (int pr = wid; pr < nv * nv; pr += 4) {
    const int u = pr / nv, v = pr - u * nv;
    if (v < u) continue;  // symmetric: filled from (v, u) below
    const int qu = s_list[u], qv = s_list[v];
    const float* Pu = ws_mean + (size_t)(p0 + qu) * J * 3;
    const float* Pv = ws_mean + (size_t)(p0 + qv) * J * 3;
    const float su = s_scale[qu], sv = s_scale[qv];
    const float msq = __fdiv_rn(__fadd_rn(su, sv), 2.0f);
    const float fu = sqrtf(__fdiv_rn(msq, su)), fv = sqrtf(__fdiv_rn(msq, sv));
    for (int j = lane; j < J; j += 64) {
      const double dx = (double)__fmul_rn(fu, Pu[j * 3]) - (double)__fmul_rn(fv, Pv[j * 3]);
      const double dy = (double)__fmul_rn(fu, Pu[j * 3 + 1]) - (double)__fmul_rn(fv, Pv[j * 3 + 1]);
      const double dz = (double)__fmul_rn(fu, Pu[j * 3 + 2]) - (double)__fmul_rn(fv, Pv[j * 3 + 2]);
      // the reference subtracts in f32 before the norm: round the differences as it does
      const float fx = (float)dx, fy = (float)dy, fz = (float)dz;
      dist[j] = (float)sqrt((double)fx * fx + (double)fy * fy + (double)fz * fz);
    }
    __builtin_amdgcn_wave_barrier();
    // mean over the k LARGEST distances of relu(1 - d / 300): rank by (d desc, index asc)
    double acc = 0.0;
    for (int j = lane; j < J; j += 64) {
      const float dj = dist[j];
      int rank = 0;
      for (int i = 0; i < J; ++i) {
        const float di = dist[i];
        rank += (di > dj) || (di == dj && i < j);
      }
      if (rank < k) acc += (double)fmaxf(__fsub_rn(1.0f, __fdiv_rn(dj, fa.sim_scale_mm)), 0.0f);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      const float sim = k > 0 ? (float)(acc / (double)k) : NAN;  // torch.mean of an empty slice
      S[u * nv + v] = sim;
      S[v * nv + u] = sim;
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // ---- 4. greedy NMS, highest score first (stable), then the output order
  for (int u = tid; u < nv; u += 256) {
    const float su = boxes[(p0 + s_list[u]) * 5 + 4];
    int rank = 0;
    for (int v = 0; v < nv; ++v) {
      const float sv = boxes[(p0 + s_list[v]) * 5 + 4];
      rank += (sv > su) || (sv == su && v < u);
    }
    s_rank[rank] = u;   // s_rank[r] = r-th best valid pose
    s_sel[u] = 0;       // 0 = undecided, 1 = kept, -1 = suppressed
  }
  __syncthreads();
  if (tid == 0) s_nsel = 0;
  __syncthreads();
  for (int r = 0; r < nv; ++r) {
    const int u = s_rank[r];
    const bool take = s_sel[u] == 0 && (!fa.order_by_score || s_nsel < fa.max_output);
    __syncthreads();
    if (!take) continue;
    if (tid == 0) {
      s_sel[u] = 1;
      s_valid[s_nsel] = u;  // (s_valid is free now: reuse as the kept list, in score order)
      s_nsel = s_nsel + 1;
    }
    for (int r2 = r + 1 + tid; r2 < nv; r2 += 256) {
      const int v = s_rank[r2];
      if (s_sel[v] == 0 && S[u * nv + v] > fa.sim_threshold) s_sel[v] = -1;
    }
    __syncthreads();
  }
  __syncthreads();
  const int nsel = s_nsel;
  if (fa.order_by_score) {
    for (int i = tid; i < n; i += 256) keep_idx[p0 + i] = i < nsel ? p0 + s_list[s_valid[i]] : -1;
  } else {  // ascending pose index: valid poses are already index-ordered in s_list
    if (tid == 0) {
      int w = 0;
      for (int u = 0; u < nv; ++u)
        if (s_sel[u] == 1) keep_idx[p0 + w++] = p0 + s_list[u];
      for (; w < n; ++w) keep_idx[p0 + w] = -1;
    }
  }
  if (tid == 0) keep_count[img] = nsel;
}

}  // namespace mtr

extern "C" size_t mtr_filter_poses_workspace_bytes(int P, int J, int max_per_image) {
  if (P < 0 || J <= 0 || max_per_image < 0) return 0;
  return ((size_t)P * J * 3 + (size_t)P * (size_t)max_per_image) * sizeof(float);
}

extern "C" int mtr_filter_poses(const float* poses3d, const float* poses2d, const float* boxes,
                                const int32_t* row_start, const int32_t* sim_offsets, int n_images,
                                int P, int A, int J, const int32_t* edges, const float* mean_bones,
                                int n_bones, int J_model, const mtr_filter_params* fp, void* workspace,
                                size_t workspace_bytes, int max_per_image, uint8_t* valid,
                                int32_t* keep_idx, int32_t* keep_count, mtr_stream_t stream) {
  if (n_images < 0 || P < 0 || A <= 0 || J <= 0 || n_bones < 0) return MTR_E_SHAPE;
  if (n_images == 0) return MTR_OK;
  if (!row_start || !sim_offsets || !keep_count || !fp) return MTR_E_NULL;
  if (P > 0 && (!poses3d || !poses2d || !boxes || !workspace || !valid || !keep_idx)) return MTR_E_NULL;
  if (n_bones > 0 && (!edges || !mean_bones)) return MTR_E_NULL;
  if (max_per_image > 1024) return MTR_E_SHAPE;  // poses per image held in LDS tables
  if (J_model <= 0 || J_model > J) return MTR_E_PARAM;
  if (workspace_bytes < mtr_filter_poses_workspace_bytes(P, J, max_per_image)) return MTR_E_WORKSPACE;
  if ((uintptr_t)workspace % 8) return MTR_E_WORKSPACE;
  mtr::FilterArgs fa;
  fa.A = A; fa.J = J; fa.J_model = J_model; fa.n_bones = n_bones;
  fa.rel_small = fp->rel_small; fa.rel_big = fp->rel_big; fa.abs_diff_mm = fp->abs_diff_mm;
  fa.stdev_mm = fp->stdev_mm; fa.box_fraction = fp->box_fraction;
  fa.sim_scale_mm = fp->sim_scale_mm; fa.sim_threshold = fp->sim_threshold;
  fa.max_output = fp->max_output; fa.var_correction = fp->var_correction;
  fa.order_by_score = fp->order_by_score;
  float* ws_mean = (float*)workspace;
  float* ws_sim = ws_mean + (size_t)P * J * 3;
  const size_t lds = (size_t)4 * ((J + 1) & ~1) * sizeof(float) + (size_t)4 * A * sizeof(double);
  if (lds > 48 * 1024) return MTR_E_SHAPE;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(mtr::pose_filter_kernel, dim3(n_images), dim3(256), lds, (hipStream_t)stream,
                     poses3d, poses2d, boxes, row_start, edges, mean_bones, fa, ws_mean, ws_sim,
                     sim_offsets, valid, keep_idx, keep_count);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}
