/*
 * ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (C99, scalar, single thread) restatement of the MeTRAbs per-crop hot path of the reference
 * (isarandi/metrabs, metrabs_pytorch/).  It is an INDEPENDENT second checker next to
 * oracle/cpu_ref.py (the torch restatement that is pinned bit-for-bit to the reference): same
 * algorithm, no torch, every function citing the reference file:line it follows.
 * tests/test_c_oracle.py pins it to the golden vectors minted from the real reference.
 *
 * Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may load this library.
 *
 * Arithmetic: float32 in the reference's op order wherever the reference is elementwise
 * (exp(x - max), scaling, warp); reductions (softmax denominator, marginals, expectations)
 * accumulate in double -- a naive sequential float loop over D*H*W terms is measurably noisier
 * (2.7e-3 mm at D=72) than torch's pairwise float reductions, and a checker should not add a
 * third source of rounding; the least-squares solve uses the double-precision normal
 * equations of the same weighted ridge system (the reference calls LAPACK gelsy in fp32,
 * ptu3d.py:100-101 -- same minimiser, documented deviation).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int proc_side, stride_train, stride_test, centered_stride, legacy_centered_stride_bug;
  int weak_perspective, mix_enabled;
  float box_size_mm, mix_3d_inside_fov;
} orc_config;

/* ptu.linspace(0, 1, n)[i] with the num==1 midpoint rule, ptu.py:78-92 */
static float lin01(int i, int n) { return n <= 1 ? 0.5f : (float)i / (float)(n - 1); }

/* models/util.py:6-20 */
static float heatmap_to_image(float c, const orc_config* f) {
  int last = f->proc_side - 1;
  float out = c * (float)(last - (last % f->stride_test));
  if (f->centered_stride) out = out + (float)(f->stride_test / 2);
  if (f->legacy_centered_stride_bug) out = out + (float)(f->stride_test / 2);
  return out;
}

/* MetrabsHeads.forward after the conv, models/metrabs.py:78-85; ptu.softmax ptu.py:47-51;
 * ptu.decode_heatmap ptu.py:58-75.  logits [B, J*(1+D), H, W]; channel J + d*J + j = slice d of j. */
void orc_decode(const float* logits, int B, int J, int D, int H, int W, const orc_config* f,
                float* coords2d, float* coords3d_rel) {
  const int HW = H * W;
  float* e = (float*)malloc(sizeof(float) * (size_t)D * HW);
  for (int b = 0; b < B; ++b) {
    const float* crop = logits + (size_t)b * J * (1 + D) * HW;
    for (int j = 0; j < J; ++j) {
      /* ---- 3D: joint softmax over (d, h, w): exp(x - amax) / sum */
      float mx = -INFINITY;
      for (int d = 0; d < D; ++d)
        for (int p = 0; p < HW; ++p) mx = fmaxf(mx, crop[(size_t)(J + d * J + j) * HW + p]);
      double sum = 0.0;
      for (int d = 0; d < D; ++d)
        for (int p = 0; p < HW; ++p) {
          e[d * HW + p] = expf(crop[(size_t)(J + d * J + j) * HW + p] - mx);
          sum += e[d * HW + p];
        }
      /* marginalise over the other two axes, then dot with linspace (x over W, y over H, z over D) */
      double ax = 0.0, ay = 0.0, az = 0.0;
      for (int w = 0; w < W; ++w) {
        double m = 0.0;
        for (int d = 0; d < D; ++d)
          for (int h = 0; h < H; ++h) m += e[d * HW + h * W + w] / sum;
        ax += m * lin01(w, W);
      }
      for (int h = 0; h < H; ++h) {
        double m = 0.0;
        for (int d = 0; d < D; ++d)
          for (int w = 0; w < W; ++w) m += e[d * HW + h * W + w] / sum;
        ay += m * lin01(h, H);
      }
      for (int d = 0; d < D; ++d) {
        double m = 0.0;
        for (int p = 0; p < HW; ++p) m += e[d * HW + p] / sum;
        az += m * lin01(d, D);
      }
      float cx = (float)ax, cy = (float)ay, cz = (float)az;
      float* o3 = coords3d_rel + ((size_t)b * J + j) * 3;
      /* heatmap_to_metric, models/util.py:29-33 */
      o3[0] = heatmap_to_image(cx, f) * f->box_size_mm / (float)f->proc_side;
      o3[1] = heatmap_to_image(cy, f) * f->box_size_mm / (float)f->proc_side;
      o3[2] = cz * f->box_size_mm;
      /* ---- 2D: softmax over (h, w) */
      const float* l2 = crop + (size_t)j * HW;
      mx = -INFINITY;
      for (int p = 0; p < HW; ++p) mx = fmaxf(mx, l2[p]);
      sum = 0.0;
      for (int p = 0; p < HW; ++p) { e[p] = expf(l2[p] - mx); sum += e[p]; }
      ax = ay = 0.0;
      for (int w = 0; w < W; ++w) {
        double m = 0.0;
        for (int h = 0; h < H; ++h) m += e[h * W + w] / sum;
        ax += m * lin01(w, W);
      }
      for (int h = 0; h < H; ++h) {
        double m = 0.0;
        for (int w = 0; w < W; ++w) m += e[h * W + w] / sum;
        ay += m * lin01(h, H);
      }
      cx = (float)ax; cy = (float)ay;
      coords2d[((size_t)b * J + j) * 2 + 0] = heatmap_to_image(cx, f);
      coords2d[((size_t)b * J + j) * 2 + 1] = heatmap_to_image(cy, f);
    }
  }
  free(e);
}

static void inv3(const double* m, double* o) {
  double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  double inv = 1.0 / (a * A + b * B + c * C);
  o[0] = A * inv; o[1] = -(b * i - c * h) * inv; o[2] = (b * f - c * e) * inv;
  o[3] = B * inv; o[4] = (a * i - c * g) * inv;  o[5] = -(a * f - c * d) * inv;
  o[6] = C * inv; o[7] = -(a * h - b * g) * inv; o[8] = (a * e - b * d) * inv;
}

/* ptu3d.is_within_fov, ptu3d.py:113-121 */
static int in_fov(float x, float y, const orc_config* f) {
  float off = f->centered_stride ? 0.f : -(float)f->stride_train / 2.f;
  float lo = (float)f->stride_train * 0.75f + off;
  float hi = (float)f->proc_side - (float)f->stride_train * 0.75f + off;
  return x >= lo && x <= hi && y >= lo && y <= hi;
}

/* ptu3d.reconstruct_ref_weakpersp, ptu3d.py:36-49 with ptu.mean_stdev_masked / reduce_mean_masked,
   ptu.py:4-34: masked means over the in-FOV joints, stdev over joints AND the two coordinates,
   depth of the reference point = stdev3d / stdev2d (both floored at 1e-5; no valid joint: the
   means are nan_to_num(0/0) = 0 and both stdevs sqrt(1e-10)) */
static void ref_weakpersp(const double* n2, const float* rel, const int* valid, int J, float* ref) {
  double nv = 0, m3[3] = {0, 0, 0}, m2[2] = {0, 0};
  for (int j = 0; j < J; ++j)
    if (valid[j]) {
      nv += 1;
      for (int k = 0; k < 3; ++k) m3[k] += rel[j * 3 + k];
      m2[0] += n2[j * 2]; m2[1] += n2[j * 2 + 1];
    }
  if (nv > 0) {
    for (int k = 0; k < 3; ++k) m3[k] /= nv;
    m2[0] /= nv; m2[1] /= nv;
  }
  double q3 = 0, q2 = 0;
  for (int j = 0; j < J; ++j)
    if (valid[j])
      for (int k = 0; k < 2; ++k) {
        double d3 = rel[j * 3 + k] - m3[k], d2 = n2[j * 2 + k] - m2[k];
        q3 += d3 * d3; q2 += d2 * d2;
      }
  double s3 = sqrt((nv > 0 ? q3 / nv : 0.0) + 1e-10), s2 = sqrt((nv > 0 ? q2 / nv : 0.0) + 1e-10);
  if (s3 < 1e-5) s3 = 1e-5;
  if (s2 < 1e-5) s2 = 1e-5;
  double z = s3 / s2;
  ref[0] = (float)(m2[0] * z - m3[0]);
  ref[1] = (float)(m2[1] * z - m3[1]);
  ref[2] = (float)(z - m3[2]);
}

/* ptu3d.reconstruct_absolute, ptu3d.py:9-33, with reconstruct_ref_fullpersp (:56-105) or
   reconstruct_ref_weakpersp (:36-49) */
int orc_reconstruct(const float* coords2d, const float* rel, const float* K, int B, int J,
                    const orc_config* f, float* out) {
  double* n2 = (double*)malloc(sizeof(double) * (size_t)B * J * 2);
  double s2d = 0.0, srb = 0.0;
  for (int b = 0; b < B; ++b) {
    double Km[9], Ki[9];
    for (int k = 0; k < 9; ++k) Km[k] = K[(size_t)b * 9 + k];
    inv3(Km, Ki);
    for (int j = 0; j < J; ++j) {
      size_t o = (size_t)b * J + j;
      double x = coords2d[o * 2], y = coords2d[o * 2 + 1];
      double nx = Ki[0] * x + Ki[1] * y + Ki[2], ny = Ki[3] * x + Ki[4] * y + Ki[5];
      n2[o * 2] = nx; n2[o * 2 + 1] = ny;
      double bx = nx * rel[o * 3 + 2] - rel[o * 3], by = ny * rel[o * 3 + 2] - rel[o * 3 + 1];
      s2d += nx * nx + ny * ny;
      srb += bx * bx + by * by;
    }
  }
  /* rms over the WHOLE batch, ptu3d.py:71-74 */
  double scale2d = sqrt(s2d / ((double)B * J * 2)), scale_rb = sqrt(srb / ((double)B * J * 2));
  for (int b = 0; b < B; ++b) {
    double m02 = 0, m12 = 0, m22 = 0, sw = 0, v0 = 0, v1 = 0, v2 = 0;
    for (int j = 0; j < J; ++j) {
      size_t o = (size_t)b * J + j;
      double xh = n2[o * 2] / scale2d, yh = n2[o * 2 + 1] / scale2d;
      double bx = (n2[o * 2] * rel[o * 3 + 2] - rel[o * 3]) / scale_rb;
      double by = (n2[o * 2 + 1] * rel[o * 3 + 2] - rel[o * 3 + 1]) / scale_rb;
      float wf = (in_fov(coords2d[o * 2], coords2d[o * 2 + 1], f) ? 1.f : 0.f) + 1e-4f;
      double w2 = (double)wf * wf;
      sw += w2; m02 -= w2 * xh; m12 -= w2 * yh; m22 += w2 * (xh * xh + yh * yh);
      v0 += w2 * bx; v1 += w2 * by; v2 -= w2 * (xh * bx + yh * by);
    }
    double l2 = (double)sqrtf(1e-2f) * (double)sqrtf(1e-2f);
    double M[9] = {sw + l2, 0, m02, 0, sw + l2, m12, m02, m12, m22 + l2}, Mi[9];
    inv3(M, Mi);
    double r0 = Mi[0] * v0 + Mi[1] * v1 + Mi[2] * v2, r1 = Mi[3] * v0 + Mi[4] * v1 + Mi[5] * v2,
           r2 = Mi[6] * v0 + Mi[7] * v1 + Mi[8] * v2;
    float ref[3] = {(float)(r0 * scale_rb), (float)(r1 * scale_rb), (float)(r2 * (scale_rb / scale2d))};
    if (f->weak_perspective) {
      int* valid = (int*)malloc(sizeof(int) * (size_t)J);
      for (int j = 0; j < J; ++j)
        valid[j] = in_fov(coords2d[((size_t)b * J + j) * 2], coords2d[((size_t)b * J + j) * 2 + 1], f);
      ref_weakpersp(n2 + (size_t)b * J * 2, rel + (size_t)b * J * 3, valid, J, ref);
      free(valid);
    }
    for (int j = 0; j < J; ++j) {
      size_t o = (size_t)b * J + j;
      float a3[3] = {rel[o * 3] + ref[0], rel[o * 3 + 1] + ref[1], rel[o * 3 + 2] + ref[2]};
      float* dst = out + o * 3;
      if (in_fov(coords2d[o * 2], coords2d[o * 2 + 1], f)) {
        float depth = rel[o * 3 + 2] + ref[2];
        float b2[3] = {(float)n2[o * 2] * depth, (float)n2[o * 2 + 1] * depth, depth};
        for (int k = 0; k < 3; ++k)
          dst[k] = f->mix_enabled ? f->mix_3d_inside_fov * a3[k] + (1.f - f->mix_3d_inside_fov) * b2[k]
                                  : b2[k];
      } else {
        for (int k = 0; k < 3; ++k) dst[k] = a3[k];
      }
    }
  }
  free(n2);
  return 0;
}

/* (u8/255)**2.2 (multiperson_model.py:196) and the 2x2 box pyramid (warping.py:10-13) */
void orc_pyramid(const uint8_t* img, int planes, int H, int W, float* l0, float* l1, float* l2) {
  for (size_t i = 0; i < (size_t)planes * H * W; ++i) l0[i] = powf((float)img[i] / 255.f, 2.2f);
  int H1 = H / 2, W1 = W / 2, H2 = H1 / 2, W2 = W1 / 2;
  for (int p = 0; p < planes; ++p) {
    for (int y = 0; y < H1; ++y)
      for (int x = 0; x < W1; ++x) {
        const float* s = l0 + (size_t)p * H * W;
        l1[((size_t)p * H1 + y) * W1 + x] =
            (((s[(2 * y) * W + 2 * x] + s[(2 * y) * W + 2 * x + 1]) + s[(2 * y + 1) * W + 2 * x]) +
             s[(2 * y + 1) * W + 2 * x + 1]) / 4.f;
      }
    for (int y = 0; y < H2; ++y)
      for (int x = 0; x < W2; ++x) {
        const float* s = l1 + (size_t)p * H1 * W1;
        l2[((size_t)p * H2 + y) * W2 + x] =
            (((s[(2 * y) * W1 + 2 * x] + s[(2 * y) * W1 + 2 * x + 1]) + s[(2 * y + 1) * W1 + 2 * x]) +
             s[(2 * y + 1) * W1 + 2 * x + 1]) / 4.f;
      }
  }
}

static float tap(const float* pl, int x, int y, int W, int H) {
  return (x >= 0 && x < W && y >= 0 && y < H) ? pl[(size_t)y * W + x] : 0.f;
}

/* warping.warp_single_image, warping.py:41-54 (+ distort_points :57-62,90-107):
 * image [3,H,W] of the chosen pyramid level, K_level [9], Hinv [9], dist [12] -> out [3,res,res] */
void orc_warp(const float* image, int H, int W, const float* Kl, const float* Hinv,
              const float* d, int res, float* out) {
  int has = 0;
  for (int k = 0; k < 12; ++k) has |= d[k] != 0.f;
  for (int v = 0; v < res; ++v)
    for (int u = 0; u < res; ++u) {
      float U = (float)u, V = (float)v;
      float ox = Hinv[0] * U + Hinv[1] * V + Hinv[2], oy = Hinv[3] * U + Hinv[4] * V + Hinv[5],
            oz = Hinv[6] * U + Hinv[7] * V + Hinv[8];
      float nx = ox / oz, ny = oy / oz;
      if (has) {
        float r2 = nx * nx + ny * ny;
        float a = (((d[4] * r2 + d[1]) * r2 + d[0]) * r2 + 1.f) / (((d[7] * r2 + d[6]) * r2 + d[5]) * r2 + 1.f);
        float b = 2.f * (nx * d[3] + ny * d[2]);
        float cx = (d[9] * r2 + d[3] + d[8]) * r2, cy = (d[11] * r2 + d[2] + d[10]) * r2;
        float sx = nx * (a + b) + cx, sy = ny * (a + b) + cy;
        nx = sx; ny = sy;
      }
      float qx = Kl[0] * nx + Kl[1] * ny + Kl[2], qy = Kl[3] * nx + Kl[4] * ny + Kl[5];
      /* normalise (warping.py:50-51) and grid_sample's align_corners=True un-normalise */
      float gx = (qx / (float)(W - 1)) * 2.f - 1.f, gy = (qy / (float)(H - 1)) * 2.f - 1.f;
      float ix = ((gx + 1.f) / 2.f) * (float)(W - 1), iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
      for (int c = 0; c < 3; ++c) out[((size_t)c * res + v) * res + u] = 0.f;
      if (!(ix > -2.f && iy > -2.f && ix < (float)W + 1.f && iy < (float)H + 1.f)) continue;
      float fx0 = floorf(ix), fy0 = floorf(iy);
      int x0 = (int)fx0, y0 = (int)fy0;
      float tx1 = ix - fx0, tx0 = (fx0 + 1.f) - ix, ty1 = iy - fy0, ty0 = (fy0 + 1.f) - iy;
      for (int c = 0; c < 3; ++c) {
        const float* pl = image + (size_t)c * H * W;
        out[((size_t)c * res + v) * res + u] =
            tap(pl, x0, y0, W, H) * (tx0 * ty0) + tap(pl, x0 + 1, y0, W, H) * (tx1 * ty0) +
            tap(pl, x0, y0 + 1, W, H) * (tx0 * ty1) + tap(pl, x0 + 1, y0 + 1, W, H) * (tx1 * ty1);
      }
    }
}
