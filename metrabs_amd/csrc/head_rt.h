// Row-tile core of the fused head for f32 features (csrc/head_rt.hip): geometry shared by the
// packer, the kernel and the dispatch in head_fused.hip.
#pragma once
#include "common.h"

namespace mtr {

// ---- row plan.  The 1x1 projection has J*(1+D) output channels: per joint one row of the 2D
// heatmap (a softmax of its own) and D rows of the volumetric heatmap (one softmax over D*H*W,
// models/metrabs.py:78-83).  A decode UNIT is therefore 1 row ("U2") or D rows ("U3").  Rows are
// re-ordered once, at pack time, into 16-row MFMA tiles so that no unit straddles a workgroup's
// block of tiles:
//   * D <= 16: an ATOM is one tile holding upt = 16 / D whole U3 units; the rows left over in an
//     atom (and the unused unit slots of the last atom) are filled with U2 rows;
//   * D  > 16: an atom is a = ceil(D / 16) consecutive tiles holding ONE U3 unit, the 16a - D rows
//     behind it filled with U2 rows;
//   * U2 rows that found no room follow in tiles of 16.
// J = 17, D = 8: 8 tiles of two joints' depth slices, one tile [joint 16's slices | 2D rows 0..7],
// one tile [2D rows 8..16] = 10 tiles = 160 rows for 153 channels (the joint-group layout of the
// first version needed 3 x 64).  J = 17, D = 72: 17 atoms of 5 tiles [72 slices | 8 2D rows].
struct RtGeom {
  int a;        // tiles per U3 atom
  int upt;      // U3 units per atom
  int n3;       // U3 atoms
  int f;        // free rows per full atom
  int u_last;   // U3 units in the last atom
  int cap2;     // U2 rows that fit into the atoms
  int n_tiles;  // tiles in all
};

constexpr int kRtMaxD = 80;  // a <= 5 tiles per atom (register budget of the 5-tile body)

__host__ __device__ inline RtGeom rt_geom(int J, int D) {
  RtGeom g;
  if (D <= 16) {
    g.a = 1;
    g.upt = 16 / D;
    g.n3 = (J + g.upt - 1) / g.upt;
  } else {
    g.a = (D + 15) / 16;
    g.upt = 1;
    g.n3 = J;
  }
  g.f = 16 * g.a - g.upt * D;
  g.u_last = J - (g.n3 - 1) * g.upt;
  g.cap2 = (g.n3 - 1) * g.f + (16 * g.a - g.u_last * D);
  const int extra = J > g.cap2 ? J - g.cap2 : 0;
  g.n_tiles = g.n3 * g.a + (extra + 15) / 16;
  return g;
}

// what packed row r holds: kind 0 = padding, 1 = U2 row of `joint`, 2 = depth slice d of `joint`
struct RtRow { int kind, joint, d; };

__host__ __device__ inline RtRow rt_row(const RtGeom& g, int J, int D, int r) {
  RtRow o{0, 0, 0};
  const int atom_rows = g.a * 16, rows3 = g.n3 * atom_rows;
  int u2;
  if (r < rows3) {
    const int atom = r / atom_rows, off = r - atom * atom_rows;
    const int nu = atom == g.n3 - 1 ? g.u_last : g.upt;
    if (off < nu * D) {
      o.kind = 2;
      o.joint = atom * g.upt + off / D;
      o.d = off % D;
      return o;
    }
    u2 = atom * g.f + (off - nu * D);
  } else {
    u2 = g.cap2 + (r - rows3);
  }
  if (u2 < J) {
    o.kind = 1;
    o.joint = u2;
  }
  return o;
}

__host__ __device__ inline int rt_encode(const RtRow& r) { return r.kind | (r.d << 2) | (r.joint << 16); }

// tiles per workgroup block: whole atoms; for one-tile atoms a launch-time choice (rtg_hint 1..5)
inline int rt_block_tiles(const RtGeom& g, int rtg_hint) {
  if (g.a > 1) return g.a;
  return rtg_hint >= 1 && rtg_hint <= 5 ? rtg_hint : 3;
}

inline bool rt_shape_ok(int C, int J, int D) {
  return C > 0 && J > 0 && J < 65536 && D > 0 && D <= kRtMaxD;
}

// section of the packed blob: weights [stage][tile][16 rows][32 ch] as the LDS image of a tile
// (16-byte channel slots XOR-swizzled per row), bias [n_tiles*16] f32, row labels [n_tiles*16] i32
inline size_t rt_section_bytes(int C, int J, int D) {
  if (!rt_shape_ok(C, J, D)) return 0;
  const RtGeom g = rt_geom(J, D);
  const size_t n_stages = (C + 31) / 32;
  return n_stages * g.n_tiles * 2048 + (size_t)g.n_tiles * 16 * 8;
}

int rt_pack(const float* weight, const float* bias, int C, int J, int D, void* section,
            hipStream_t stream);

// features f32 [B,C,H,W] / [B,H,W,C]; H*W % 4 == 0, NHWC: C % 4 == 0 (checked by the caller).
// rtg_hint: tiles per workgroup for one-tile atoms (0 = pick from the launch size); np_hint: column
// blocks of 64 positions per workgroup tile for maps of more than 64 positions (0 = pick, 1 = one
// K loop per column block, 2..4); ks_hint: K groups per workgroup, 0 = pick, 1 = one,
// 2 = two wherever the kernel can (C % 64 == 0, blocks of <= 3 tiles).
// ld_hint: loader-wave kernel (four MFMA waves + one wave that issues every copy), 0 = pick, 1 =
// never, 2 = whenever it can (C % 32 == 0); split_hint: the column blocks of a map of more than 64
// positions dealt to different workgroups and merged by a second, tiny launch through `workspace`
// (rt_workspace_bytes; may be NULL: no split), 0 = pick, 1 = never, 2 = whenever possible.
int rt_launch(const float* feat, int layout, const void* section, int B, int C, int H, int W, int J,
              int D, const HeadScale& hs, float* coords2d, float* coords3d_rel, int rtg_hint, int np_hint,
              int ks_hint, int ld_hint, int split_hint, void* workspace, size_t workspace_bytes,
              hipStream_t stream);
size_t rt_workspace_bytes(int B, int J, int D, int H, int W);

// which kernel a launch takes: kernel id (MTR_HEAD_KERNEL_* of the header), tiles per workgroup, column
// blocks per workgroup tile (np kernel), column-block split (0 = none), workgroups of the main launch
enum { kRtKernelPlain = 1, kRtKernelLoader = 2, kRtKernelTwoKGroups = 3, kRtKernelNp = 4, kRtKernel16 = 12 };
struct RtDispatch { int kernel, rtg, np, split; long long n_wg; double model_us; int pack; };  // model_us: the plan's estimate (0: none); pack: crops per packed last block
RtDispatch rt_dispatch(int B, int C, int H, int W, int J, int D, int rtg_hint, int np_hint, int ks_hint,
                       int ld_hint, int split_hint, bool have_workspace);

// ---- 16-bit features on the row-tile core (head_rt16_kernel): C % 64 == 0, D <= 80, any map size; the
// section of the packed blob: [stage of 64 ch][tile][16 rows][64 ch] in the feature dtype + bias + labels
size_t rt16_section_bytes(int C, int J, int D);
int rt16_pack(const float* weight, const float* bias, int C, int J, int D, int feat_dtype, void* section,
              hipStream_t stream);
size_t rt16_workspace_bytes(int B, int C, int J, int D, int H, int W, int layout);
RtDispatch rt16_dispatch(int B, int H, int W, int J, int D, int rtg_hint, int split_hint, bool have_split_ws);
int rt16_launch(const void* feat, int feat_dtype, int layout, const void* section, int B, int C, int H, int W,
                int J, int D, const HeadScale& hs, float* coords2d, float* coords3d_rel, int rtg_hint,
                int split_hint, void* workspace, size_t workspace_bytes, hipStream_t stream);

}  // namespace mtr
