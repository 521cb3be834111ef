// aten's separable bilinear resize (antialiased when shrinking), restated for the device:
// aten/native/cpu/UpSampleKernel.cpp.  Shared by K9 (detector pre-processing, detector_pre.hip) and
// the antialias_factor > 4 branch of the crop sampler (warp.hip), which both replace
// torchvision.transforms.functional.resize(..., antialias=True) = F.interpolate(mode='bilinear',
// antialias=True) (person_detector.py:23-24, multiperson_model.py:312-315).
#pragma once
#include "common.h"

namespace mtr {

constexpr int kDTaps = 40;   // taps per output index: ceil(2 * scale) + 2 <= 40 (scale <= 19)

struct AxisGeom {
  int in_size, out_size;  // frame / resized extent along this axis
  int aa;                 // antialias (shrinking) or plain bilinear
};

// aten/native/cpu/UpSampleKernel.cpp: _compute_indices_min_size_weights_aa (antialias) and
// compute_indices_weights / guard_index_and_lambda (plain).  Split in two so that the taps can be
// evaluated in parallel: axis_span (per output index) and axis_raw_weight (per tap).
struct AxisSpan {
  int imin, isize;
  float center, invscale, l1;  // l1: plain-bilinear lambda
};
__device__ __forceinline__ AxisSpan axis_span(int i, const AxisGeom& g) {
  AxisSpan s;
  const float scale = (float)g.in_size / (float)g.out_size;
  if (g.aa) {
    const float support = scale >= 1.0f ? scale : 1.0f;
    s.invscale = scale >= 1.0f ? (float)(1.0 / (double)scale) : 1.0f;
    s.center = (float)((double)scale * ((double)i + 0.5));
    long long lo = (long long)((double)(s.center - support) + 0.5);
    long long hi = (long long)((double)(s.center + support) + 0.5);
    if (lo < 0) lo = 0;
    if (hi > g.in_size) hi = g.in_size;
    s.imin = (int)lo;
    s.isize = min((int)(hi - lo), kDTaps);  // (host rejects scales that need more)
    s.l1 = 0.0f;
  } else {
    float real = (float)((double)scale * ((double)i + 0.5) - 0.5);
    if (real < 0.0f) real = 0.0f;
    int i0 = (int)floorf(real);
    if (i0 > g.in_size - 1) i0 = g.in_size - 1;
    s.imin = i0;
    s.isize = 2;  // tap 1 is read at min(i0 + 1, in_size - 1)
    s.l1 = fminf(fmaxf(real - (float)i0, 0.0f), 1.0f);
    s.center = s.invscale = 0.0f;
  }
  return s;
}
// tap j before normalisation (antialias) / final weight (plain bilinear)
__device__ __forceinline__ float axis_raw_weight(const AxisSpan& s, int j, int aa) {
  if (!aa) return j == 0 ? 1.0f - s.l1 : s.l1;
  float x = (float)(((double)((float)(j + s.imin) - s.center) + 0.5) * (double)s.invscale);
  x = x < 0.0f ? -x : x;
  return x < 1.0f ? 1.0f - x : 0.0f;
}


}  // namespace mtr
