// K1 + K2-K4 fused, f32 features: row-tile core.
//
// Replaces MetrabsHeads.forward (metrabs_pytorch/models/metrabs.py:75-85): the 1x1 projection
// logits[n, p] = sum_c W[n, c] * feat[c, p] + bias[n] on the matrix cores with the two soft-argmax
// decodes (ptu.py:47-75) and the scaling (models/util.py:6-33) as the epilogue; the logits never
// leave the CU.
//
// Decomposition
//   * rows (output channels) are packed into 16-row tiles so that every decode unit -- one 2D
//     heatmap row, or the D depth slices of one joint -- lies inside one workgroup's block of
//     tiles (head_rt.h); 153 channels at J = 17, D = 8 are 10 tiles (160 rows);
//   * workgroup = (crop, block of RT <= 5 consecutive tiles), 4 waves; wave w owns the 16 positions
//     16 w .. 16 w + 15 of a 64-position column block and ALL RT row tiles: RT independent
//     v_mfma_f32_16x16x4_f32 accumulators, so consecutive MFMAs never depend on each other;
//   * K streams in 32-channel stages through a ring of LDS buffers filled by global_load_lds
//     (no staging registers, no ds_write pass): the weights are packed as the LDS image of their
//     tile, NHWC features land K-contiguous with the swizzle applied on the source side, NCHW
//     features keep their [channel][position] layout and are read one float per MFMA.  The DMA is
//     issued from inline asm with a counted s_waitcnt vmcnt, because the compiler makes every
//     ds_read that may alias a builtin global_load_lds wait for vmcnt(0), i.e. prefetch depth 1;
//   * arithmetic: exact-f32 MFMA chains of 16 channels; the two chains of a stage are added in f32
//     and carried into f64 accumulators on the VALU underneath the next stage's MFMAs
//     (tools/experiments/carry_scheme_sim.py: 7e-4 mm from the fp64 truth on the peaked golden
//     case where the reference's own oneDNN conv is 2.7e-3 mm; 2.4e-4 = 1 ulp on the others);
//   * maps of more than 64 positions run one K loop per 64-position column block and merge the
//     blocks' (max, sums) like an online softmax, so any H*W fits the same registers;
//   * epilogue: logits (+bias) -> LDS, a 16-lane group per row: row max -> unit max -> f64
//     exp / moment sums per row -> one thread per unit adds its rows and writes the coordinates.
//
// Round 3
//   * head_rt_ld_kernel: a fifth wave (the loader) issues every copy of a stage and waits for it; the
//     four MFMA waves only meet it at the stage barrier (rt_loader_loop, rt_block<..., LD = true>);
//   * the column blocks of a map of more than 64 positions can be dealt to different workgroups, their
//     softmax statistics merged by head_rt_merge_kernel through a caller-provided workspace
//     (mtr_head_fused_ws) -- bit-identical to one workgroup walking them;
//   * the launch plan (which kernel, how many tiles per workgroup, split or not) minimises a measured
//     cost model (rt_plan; host-visible through mtr_head_plan);
//   * head_rt16_kernel: the same row plan, LDS images, loader wave and decode for f16 / bf16 features on
//     v_mfma_f32_16x16x32 (the shapes the joint-group kernels of head_fused.hip do not take).
#include "head_rt.h"

#include <array>
#include <map>
#include <mutex>
#include <queue>
#include <tuple>
#include <type_traits>
#include <vector>

namespace mtr {

using v4f = __attribute__((ext_vector_type(4))) float;

// developer-only timing ablations (tools/experiments/ablate_rt.py); 0 in the product:
// 1 = no decode epilogue, 2 = no MFMA, 4 = no copies inside the K loop, 8 = no f64 carry (one f32
// chain per tile over all of K), 16 = no fragment reads
#ifndef MTR_RT_ABLATE
#define MTR_RT_ABLATE 0
#endif
#ifndef MTR_RT_NBUF
#define MTR_RT_NBUF 0       // 0: by block size (rt_nbuf)
#endif
#ifndef MTR_RT_PAIR
#define MTR_RT_PAIR 1       // 1: 4-slot rings are filled two stages ahead and the workgroup meets at a barrier
                            // every SECOND stage (one K group only)
#endif
#ifndef MTR_RT_KS_NBUF
#define MTR_RT_KS_NBUF 2    // ring depth of each K group of head_rt_ks_kernel
#endif
#ifndef MTR_RT_SPREAD_READS
#define MTR_RT_SPREAD_READS 0   // 1: the fragment reads of a half stage go one per MFMA shadow of the half stage in
                                // front of it instead of as a burst of RT + NP ds_reads (the four MFMA waves of a
                                // workgroup run in lock-step: their bursts meet in the LDS queue and no MFMA issues
                                // behind a queued read)
#endif
#ifndef MTR_RT_SCALAR_ADD
#define MTR_RT_SCALAR_ADD 0
#endif
#ifndef MTR_RT_LD_NBUF
#define MTR_RT_LD_NBUF 4    // ring depth of the loader-wave kernels (head_rt_ld_kernel) for blocks of <= 3 tiles
#endif
#ifndef MTR_RT_LD_LA
#define MTR_RT_LD_LA 3      // ... and the stages its loader runs ahead (<= ring depth - 1; copies per stage x (LA - 1)
                            // must fit the 6-bit vmcnt)
#endif
#ifndef MTR_RT_DECODE_PER_ROW
#define MTR_RT_DECODE_PER_ROW 0   // 1: rounds 1-3's decode epilogue (a 16-lane group per row, three passes, two
                                  // barriers, one thread per unit for the last sums); 0: a WAVE per softmax unit
                                  // (round 4, rt_decode_units)
#endif
#ifndef MTR_RT_EXP32
#define MTR_RT_EXP32 1      // 1: v_exp_f32 in the decode epilogue (f64 sums; measured -0.9 us at B=64, -11 us at
                            // B=1024, parity unchanged: the stand-alone decode does the same); 0: f64 polynomial
#endif

// LDS ring: NBUF - 1 stages in flight behind the one being consumed.  Blocks of <= 3 tiles (the
// small-launch configuration: one workgroup per CU, nothing else to hide the first HBM misses) keep
// 3 stages in flight; 4- and 5-tile blocks run with one stage in flight and two workgroups per CU
// (the other workgroup covers the latency; measured equal to the deep ring at every launch size,
// tools/experiments/ablate_rt.py) so that 5 tiles' accumulators and their LDS fit twice.  Tiles of
// several column blocks: the deep ring while it fits the 160 KiB.
constexpr int kRtLP = 68;         // logits row pitch in LDS (floats) per 64-position column block
constexpr int kRtChunkNCHW = 1088;  // 4 channel rows of 64 positions + 64 B: the two channel groups
                                    // a 32-lane ds_read_b32 group touches land 16 banks apart

// A workgroup's tile is RT row tiles (16 output channels each) x NP column blocks (64 positions
// each): one K loop with RT * NP accumulators per wave (wave w: positions 16 w .. 16 w + 15 of every
// column block).  NP > 1 is for maps of more than 64 positions on launches that do not fill the
// chip with multi-tile blocks: 8 * RT * NP MFMAs per stage and wave from RT + NP fragment reads.
__host__ __device__ constexpr int rt_feat_chunk(bool nhwc) { return nhwc ? 8192 : 8 * kRtChunkNCHW; }
__host__ __device__ constexpr int rt_stage_bytes(int rt, int np, bool nhwc) {
  return rt * 2048 + np * rt_feat_chunk(nhwc);
}
// logits [R][NP * 68] + per row: max, unit max, bias (f32), label (i32), 3 f64 sums, and the unit's
// running (max, 4 sums) across column blocks (5 f64)
__host__ __device__ constexpr int rt_epilogue_bytes(int rtmax, int np) {
  return rtmax * 16 * (np * kRtLP * 4 + 16 + 24 + 40);
}
__host__ __device__ constexpr int rt_nbuf(int rtmax, int np, bool nhwc) {
  if (MTR_RT_NBUF && rtmax <= 3 && np == 1) return MTR_RT_NBUF;  // (developer override: the small-launch configuration)
  if (rtmax * np > 4 || (np == 1 && rtmax > 3)) return 2;
  return 4 * rt_stage_bytes(rtmax, np, nhwc) + rt_epilogue_bytes(rtmax, np) <= 160 * 1024 ? 4 : 2;
}
__host__ __device__ constexpr int rt_lds_bytes(int rtmax, int np, bool nhwc) {
  return rt_nbuf(rtmax, np, nhwc) * rt_stage_bytes(rtmax, np, nhwc) + rt_epilogue_bytes(rtmax, np);
}

__global__ void head_rt_pack_kernel(const float* __restrict__ w, const float* __restrict__ bias, int C,
                                    int J, int D, RtGeom g, int n_stages, char* __restrict__ section) {
  float* wt = reinterpret_cast<float*>(section);
  const size_t n_w = (size_t)n_stages * g.n_tiles * 512;
  float* bias_p = wt + n_w;
  int* info = reinterpret_cast<int*>(bias_p + g.n_tiles * 16);
  const size_t total = n_w + (size_t)g.n_tiles * 16;
  for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total;
       u += (size_t)gridDim.x * blockDim.x) {
    if (u < n_w) {
      const int e = (int)(u & 3), slotp = (int)((u >> 2) & 7), row = (int)((u >> 5) & 15);
      const size_t ts = u >> 9;  // stage * n_tiles + tile
      const int tile = (int)(ts % g.n_tiles), stage = (int)(ts / g.n_tiles);
      const int c = stage * 32 + ((slotp ^ ((row >> 1) & 7)) << 2) + e;
      const RtRow rr = rt_row(g, J, D, tile * 16 + row);
      float v = 0.0f;
      if (rr.kind && c < C) v = w[(size_t)(rr.kind == 1 ? rr.joint : J + rr.d * J + rr.joint) * C + c];
      wt[u] = v;
    } else {
      const int r = (int)(u - n_w);
      const RtRow rr = rt_row(g, J, D, r);
      bias_p[r] = rr.kind ? bias[rr.kind == 1 ? rr.joint : J + rr.d * J + rr.joint] : 0.0f;
      info[r] = rt_encode(rr);
    }
  }
}

// LDS byte address of a generic pointer into the workgroup's dynamic LDS
__device__ __forceinline__ unsigned rt_lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// One wave-wide 1 KiB copy global -> LDS: lane L reads 16 bytes at sbase + voff and they land at
// lds_addr + 16 L.  sbase / lds_addr are wave-uniform (SGPRs), voff per lane.
__device__ __forceinline__ void rt_dma16(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :
               : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(sbase)
               : "memory", "m0");  // (M0 is overwritten: the register allocator must know)
}

template <int N>
__device__ __forceinline__ void rt_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

struct RtArgs {
  const float* feat;
  const char* wt;        // [n_stages][n_tiles][2048 B]
  const float* bias_p;   // [n_tiles * 16]
  const int* info;       // [n_tiles * 16]
  int B, C, H, W, J, D;
  int n_tiles, rtg, n_blocks, n_stages;
  HeadScale hs;
  AxisInv inv;           // 1 / (W - 1), 1 / (H - 1), 1 / (D - 1)
  float* c2d;
  float* c3d;
  // column blocks of a map dealt to different workgroups (maps of more than 64 positions on small
  // launches): cb_split = their number (0 = every workgroup walks all of its column blocks itself),
  // ws = [B][cb_split][n_tiles * 16][5] f64: per packed row that starts a softmax unit, the unit's
  // (max, sum e, sum e x, sum e y, sum e z) over ONE column block, merged by head_rt_merge_kernel
  double* ws;
  int cb_split;
  // cb_split launches of maps whose last column block holds 16 (32) positions (H W % 64): the last
  // blocks of pack_g = 4 (2) consecutive crops share ONE workgroup -- columns 16 s .. 16 s + 15 (32 s ..)
  // of its tile are crop c0 + s's tail -- instead of one workgroup each with 3/4 (1/2) of its columns
  // padding (12x12 maps: 2.25 instead of 3 blocks per crop).  0 = every block is one crop's.
  int pack_g;
};

constexpr int kRtCarry = 8;  // stages (of 32 channels) summed in f32 before the sum goes into f64

template <int RT, int NP>
struct RtRegs {
  double acc[RT * NP][4];   // f64 totals; accumulator q = (row tile q % RT, column block q / RT)
  v4f run[RT * NP];         // f32 sum of the finished chains of up to kRtCarry stages
  v4f part[2][RT * NP];     // the 32-channel MFMA chain of a stage, by stage parity
  v4f ya[RT], yb[NP];       // fragments of the previous stage's second half (consumed one barrier late)
};

template <int RT, int NP, bool NHWC>
__device__ __forceinline__ void rt_read_frags(const char* buf, int a_addr, int b_addr, v4f (&fa)[RT],
                                              v4f (&fb)[NP]) {
  if (MTR_RT_ABLATE & 16) {
#pragma unroll
    for (int t = 0; t < RT; ++t) fa[t] = v4f{(float)a_addr, 1.f, 2.f, (float)t};
#pragma unroll
    for (int p = 0; p < NP; ++p) fb[p] = v4f{(float)b_addr, 1.f, 2.f, (float)p};
    return;
  }
#pragma unroll
  for (int t = 0; t < RT; ++t) fa[t] = *reinterpret_cast<const v4f*>(buf + t * 2048 + a_addr);
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if constexpr (NHWC) {
      fb[p] = *reinterpret_cast<const v4f*>(buf + RT * 2048 + p * rt_feat_chunk(true) + b_addr);
    } else {
      const float* q = reinterpret_cast<const float*>(buf + RT * 2048 + p * rt_feat_chunk(false) + b_addr);
      fb[p] = v4f{q[0], q[64], q[128], q[192]};  // channels k .. k + 3 of this lane's position
    }
  }
}

// ... one of them: unit u < RT is the fragment of row tile u, unit RT + p that of column block p
template <int RT, int NP, bool NHWC>
__device__ __forceinline__ void rt_read_frag_unit(const char* buf, int a_addr, int b_addr, v4f (&fa)[RT],
                                                  v4f (&fb)[NP], int u) {
  if (MTR_RT_ABLATE & 16) {
    if (u < RT) fa[u] = v4f{(float)a_addr, 1.f, 2.f, (float)u};
    else fb[u - RT] = v4f{(float)b_addr, 1.f, 2.f, (float)u};
    return;
  }
  if (u < RT) {
    fa[u] = *reinterpret_cast<const v4f*>(buf + u * 2048 + a_addr);
  } else {
    const int p = u - RT;
    if constexpr (NHWC) {
      fb[p] = *reinterpret_cast<const v4f*>(buf + RT * 2048 + p * rt_feat_chunk(true) + b_addr);
    } else {
      const float* q = reinterpret_cast<const float*>(buf + RT * 2048 + p * rt_feat_chunk(false) + b_addr);
      fb[p] = v4f{q[0], q[64], q[128], q[192]};
    }
  }
}

// One MFMA slot: slot n of a half stage is accumulator n % (RT NP), k-step n / (RT NP)
// (accumulator-major inside a k-step, so consecutive MFMAs never share an accumulator).  A stage's
// chain starts from zero in the first k-step of its first half.
template <int RT, int NP>
__device__ __forceinline__ void rt_mfma_slot(v4f (&chain)[RT * NP], const v4f (&fa)[RT],
                                             const v4f (&fb)[NP], int n, bool first_half) {
  const int q = n % (RT * NP), k = n / (RT * NP);
  const int t = q % RT, p = q / RT;
  if (MTR_RT_ABLATE & 2) {
    chain[q][0] += fa[t][k] * fb[p][k];
    return;
  }
  chain[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(
      fa[t][k], fb[p][k],
      (first_half && k == 0 && !(MTR_RT_ABLATE & 8)) ? v4f{0.f, 0.f, 0.f, 0.f} : chain[q], 0, 0, 0);
}
// element pair e2 (two adjacent registers of tile e2 / 2) of a finished chain into the f32 running sum
template <int RT>
__device__ __forceinline__ void rt_run_add(v4f (&run)[RT], const v4f (&done)[RT], int e2) {
  if (MTR_RT_ABLATE & 8) return;
  const int t = e2 >> 1, r = (e2 & 1) * 2;
#if MTR_RT_SCALAR_ADD
  // two v_add_f32 instead of the v_pk_add_f32 the compiler forms from the pair (MI355X_MICROARCH.md:
  // packed f32 VALU beside MFMAs costs ~+13 cycles per instruction over its issue slot)
  float x0 = run[t][r], x1 = run[t][r + 1];
  asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(done[t][r]));
  asm volatile("v_add_f32 %0, %0, %1" : "+v"(x1) : "v"(done[t][r + 1]));
  run[t][r] = x0;
  run[t][r + 1] = x1;
#else
  run[t][r] += done[t][r];
  run[t][r + 1] += done[t][r + 1];
#endif
}
// the running sums into f64, and restart them
template <int RT>
__device__ __forceinline__ void rt_flush(double (&acc)[RT][4], v4f (&run)[RT]) {
#pragma unroll
  for (int t = 0; t < RT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] += (double)run[t][r];
    run[t] = v4f{0.f, 0.f, 0.f, 0.f};
  }
}

// A softmax unit's (max, sums) over column block cb are complete.  One workgroup walking all blocks:
// merge with the blocks before it like an online softmax -- the running (max, 4 sums) live in LDS
// (runstat), not in registers that would stay allocated through the K loop -- and write the
// coordinates behind the last block.  Blocks dealt to different workgroups (a.ws): hand the five
// numbers to head_rt_merge_kernel, which merges them in the same order with the same arithmetic.
__device__ __forceinline__ void rt_unit_finish(const RtArgs& a, int crop, int t0, int row, int kind, int j,
                                               int cb, int n_cb, float umax, double S, double SX, double SY,
                                               double SZ, double* runstat) {
  double run_m = (double)umax, run_s = S, run_x = SX, run_y = SY, run_z = SZ;
  if (a.cb_split) {
    double* w = a.ws + (((size_t)crop * n_cb + cb) * ((size_t)a.n_tiles * 16) + (size_t)(t0 * 16 + row)) * 5;
    w[0] = run_m; w[1] = run_s; w[2] = run_x; w[3] = run_y; w[4] = run_z;
    return;
  }
  if (cb > 0) {
    const double pm = runstat[row * 5];
    const double mn = fmax(pm, run_m);
    const double zero = pm - pm;  // 0.0 at run time (the maxima are finite)
    const double f1 = exp_neg64_late(pm - mn, zero), f2 = exp_neg64_late(run_m - mn, zero);
    run_s = runstat[row * 5 + 1] * f1 + S * f2;
    run_x = runstat[row * 5 + 2] * f1 + SX * f2;
    run_y = runstat[row * 5 + 3] * f1 + SY * f2;
    run_z = runstat[row * 5 + 4] * f1 + SZ * f2;
    run_m = mn;
  }
  if (cb < n_cb - 1) {
    runstat[row * 5] = run_m;
    runstat[row * 5 + 1] = run_s;
    runstat[row * 5 + 2] = run_x;
    runstat[row * 5 + 3] = run_y;
    runstat[row * 5 + 4] = run_z;
  }
  if (cb == n_cb - 1) {
    const size_t o = (size_t)crop * a.J + j;
    const double inv_s = fast_rcp64(run_s);
    if (kind == 1) {
      a.c2d[o * 2 + 0] = heatmap_to_px(axis_coord_rcp(run_x, inv_s, a.inv.w), a.hs);
      a.c2d[o * 2 + 1] = heatmap_to_px(axis_coord_rcp(run_y, inv_s, a.inv.h), a.hs);
    } else {
      a.c3d[o * 3 + 0] = heatmap_to_mm_xy(axis_coord_rcp(run_x, inv_s, a.inv.w), a.hs);
      a.c3d[o * 3 + 1] = heatmap_to_mm_xy(axis_coord_rcp(run_y, inv_s, a.inv.h), a.hs);
      a.c3d[o * 3 + 2] = heatmap_to_mm_z(axis_coord_rcp(run_z, inv_s, a.inv.d), a.hs);
    }
  }
}

// Column blocks dealt to different workgroups: one thread per (crop, packed row that starts a
// unit) merges the blocks' (max, sums) in block order -- the arithmetic of rt_unit_finish's LDS
// merge, so the coordinates are those of the one-workgroup walk bit for bit.
__global__ __launch_bounds__(256) void head_rt_merge_kernel(RtArgs a, int n_cb) {
  const int n_rows = a.n_tiles * 16;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)a.B * n_rows) return;
  const int crop = (int)(idx / n_rows), row = (int)(idx - (long long)crop * n_rows);
  const unsigned inf = (unsigned)a.info[row];
  const int kind = inf & 3, d = (inf >> 2) & 0x3fff, j = (int)(inf >> 16);
  if (!(kind == 1 || (kind == 2 && d == 0))) return;
  const double* w = a.ws + ((size_t)crop * n_cb * n_rows + row) * 5;
  double run_m = w[0], run_s = w[1], run_x = w[2], run_y = w[3], run_z = w[4];
  for (int cb = 1; cb < n_cb; ++cb) {
    const double* v = w + (size_t)cb * n_rows * 5;
    const double mn = fmax(run_m, v[0]);
    const double f1 = exp_neg64(run_m - mn), f2 = exp_neg64(v[0] - mn);
    run_s = run_s * f1 + v[1] * f2;
    run_x = run_x * f1 + v[2] * f2;
    run_y = run_y * f1 + v[3] * f2;
    run_z = run_z * f1 + v[4] * f2;
    run_m = mn;
  }
  const size_t o = (size_t)crop * a.J + j;
  const double inv_s = fast_rcp64(run_s);
  if (kind == 1) {
    a.c2d[o * 2 + 0] = heatmap_to_px(axis_coord_rcp(run_x, inv_s, a.inv.w), a.hs);
    a.c2d[o * 2 + 1] = heatmap_to_px(axis_coord_rcp(run_y, inv_s, a.inv.h), a.hs);
  } else {
    a.c3d[o * 3 + 0] = heatmap_to_mm_xy(axis_coord_rcp(run_x, inv_s, a.inv.w), a.hs);
    a.c3d[o * 3 + 1] = heatmap_to_mm_xy(axis_coord_rcp(run_y, inv_s, a.inv.h), a.hs);
    a.c3d[o * 3 + 2] = heatmap_to_mm_z(axis_coord_rcp(run_z, inv_s, a.inv.d), a.hs);
  }
}

// The decode of a workgroup's logits tile (rows x NP column blocks of 64 positions, row pitch NP * 68
// floats in LDS): per row the maximum, per softmax unit (a 2D row; the D depth slices of a joint) the
// unit maximum, the f64 sums of e, e x, e y per row, the unit's sums and the hand-over to
// rt_unit_finish.  A 16-lane group per row, NG groups; idle_wave: a wave that only keeps the barrier
// count (the loader wave).  Shared by the f32 kernels (rt_block) and the 16-bit one (rt16_block).
// `segs` = 4 (2): a packed last block (RtArgs::pack_g) -- quad s (half s) of a row's 16 lanes holds the 16 (32)
// positions of crop + s, the butterflies stop after their two quad steps (+ one pairing step) and every per-row number exists once
// per segment ([row * 4 + s]).  The remaining steps of the 16-lane butterflies of an unpacked last
// block only add zeros (take the maximum with -inf): the segment's numbers are those bit for bit.
// (segs = 2, a last block of 32 positions: two quads per segment, their sums paired by row_half_mirror --
// one addition, as the two rotate steps of the unpacked block come to once the other two quads are 0.)
template <typename T>
__device__ __forceinline__ T rt_seg_sum(T v, int segs) {
  v += dpp_move<kDppXor1>(v);
  v += dpp_move<kDppXor2>(v);
  if (segs == 2) {
    v += dpp_move<kDppHalfMirror>(v);
  } else if (segs == 1) {
    v += dpp_move<kDppRor4>(v);
    v += dpp_move<kDppRor8>(v);
  }
  return v;
}
__device__ __forceinline__ float rt_seg_max(float v, int segs) {
  v = fmaxf(v, dpp_move<kDppXor1>(v));
  v = fmaxf(v, dpp_move<kDppXor2>(v));
  if (segs == 2) {
    v = fmaxf(v, dpp_move<kDppHalfMirror>(v));
  } else if (segs == 1) {
    v = fmaxf(v, dpp_move<kDppRor4>(v));
    v = fmaxf(v, dpp_move<kDppRor8>(v));
  }
  return v;
}
template <int RT, int NP, int NG>
__device__ __forceinline__ void rt_decode_rows(const RtArgs& a, float* Ls, float* rowmax, float* unitmax,
                                                 int* info_s, double* rowsum, double* runstat, int tid,
                                                 bool idle_wave, int HW, int crop, int t0, int cb0, int n_cb,
                                                 int segs = 1) {
  constexpr int R = RT * 16, KR = (R + NG - 1) / NG, LP = NP * kRtLP;
#pragma unroll 1
   for (int np = 0; np < NP; ++np) {  // decode the group's column blocks one after the other
    const int cb = NP == 1 ? cb0 : cb0 + np;
    if (NP > 1 && cb >= n_cb) break;
    const float* Lb = Ls + np * kRtLP;  // this column block's 64 columns (row pitch LP)
    // ---- decode, a 16-lane group per row (RT rounds of 16 rows)
    // (the addresses below do not depend on the column block; the empty asm keeps the compiler from
    //  computing them once in front of the K loop and holding them in registers through it)
    int tid_d = tid;
    asm volatile("" : "+v"(tid_d));
    const int grp = tid_d >> 4, l16 = tid_d & 15;
    const int seg_lanes = 16 / segs;  // 16, 8 or 4 lanes of the row's group per segment
    const int seg = l16 / seg_lanes, li = l16 - seg * seg_lanes;
    const int pbase = cb * 64 + li * 4;  // this lane's 4 positions (of its segment's crop)
    v4f x[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int row = k * NG + grp;
      if ((KR * NG > R && row >= R) || idle_wave) continue;
      x[k] = *reinterpret_cast<const v4f*>(Lb + row * LP + l16 * 4);
      float m = -INFINITY;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (pbase + q < HW) m = fmaxf(m, x[k][q]);
      m = rt_seg_max(m, segs);
      if (li == 0) rowmax[row * segs + seg] = m;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int row = k * NG + grp;
      if ((KR * NG > R && row >= R) || idle_wave) continue;
      const unsigned inf = (unsigned)info_s[row];
      const int kind = inf & 3, d = (inf >> 2) & 0x3fff;
      const int first = kind == 2 ? row - d : row, n = kind == 2 ? a.D : 1;
      float m = -INFINITY;
      for (int kk = li; kk < n; kk += seg_lanes) m = fmaxf(m, rowmax[(first + kk) * segs + seg]);
      m = rt_seg_max(m, segs);
      const float nm = -m * kLog2e;
      double s = 0, sx = 0, sy = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = pbase + q;
        if (p < HW && kind != 0) {
          const double e = MTR_RT_EXP32 ? (double)exp_shifted(x[k][q], nm)
                                        : exp_neg64((double)x[k][q] - (double)m);
          const int h = p / a.W, w = p - h * a.W;
          s += e;
          sx += e * (double)w;
          sy += e * (double)h;
        }
      }
      s = rt_seg_sum(s, segs);
      sx = rt_seg_sum(sx, segs);
      sy = rt_seg_sum(sy, segs);
      if (li == 0) {
        const int o = row * segs + seg;
        rowsum[o * 3 + 0] = s;
        rowsum[o * 3 + 1] = sx;
        rowsum[o * 3 + 2] = sy;
        unitmax[o] = m;
      }
    }
    __syncthreads();
    if (a.D > 16) {  // (wave-uniform) long units: 72 depth slices
      // a unit's rows are added by the 16-lane group of its FIRST row (lane l takes rows l, l + 16,
      // ...: 5 rows per lane instead of a 72-step chain in one thread; measured 170 -> 161 us at
      // B = 64, D = 72.  Short units keep the one-thread loop below: 8 rows, and 1 - 4 % faster)
      for (int sg = 0; sg < segs; ++sg) {  // (a packed block: once per crop)
        if (crop + sg >= a.B) break;
  #pragma unroll
        for (int k = 0; k < KR; ++k) {
          const int row = k * NG + grp;
          if ((KR * NG > R && row >= R) || idle_wave) continue;
          const unsigned inf = (unsigned)info_s[row];
          const int kind = inf & 3, d = (inf >> 2) & 0x3fff, j = (int)(inf >> 16);
          if (!(kind == 1 || (kind == 2 && d == 0))) continue;  // (uniform in the group)
          const int n = kind == 2 ? a.D : 1;
          double S = 0, SX = 0, SY = 0, SZ = 0;
          for (int kk = l16; kk < n; kk += 16) {
            const int o = (row + kk) * segs + sg;
            const double s = rowsum[o * 3];
            S += s;
            SX += rowsum[o * 3 + 1];
            SY += rowsum[o * 3 + 2];
            SZ += s * (double)kk;
          }
          if (n > 1) {
            S = group_sum<16>(S);
            SX = group_sum<16>(SX);
            SY = group_sum<16>(SY);
            SZ = group_sum<16>(SZ);
          }
          if (l16 != 0) continue;
          rt_unit_finish(a, crop + sg, t0, row, kind, j, cb, n_cb, unitmax[row * segs + sg], S, SX, SY, SZ, runstat);
        }
      }
    } else {
      int tid_c = tid;
      asm volatile("" : "+v"(tid_c));
      for (int u = tid_c; u < R * segs; u += NG * 16) {  // one thread per (row, segment)
        const int row = u / segs, sg = u - row * segs;
        if (idle_wave || crop + sg >= a.B) continue;
        const unsigned inf = (unsigned)info_s[row];
        const int kind = inf & 3, d = (inf >> 2) & 0x3fff, j = (int)(inf >> 16);
        if (kind == 1 || (kind == 2 && d == 0)) {
          const int n = kind == 2 ? a.D : 1;
          double S = 0, SX = 0, SY = 0, SZ = 0;
          for (int k = 0; k < n; ++k) {
            const int o = (row + k) * segs + sg;
            const double s = rowsum[o * 3];
            S += s;
            SX += rowsum[o * 3 + 1];
            SY += rowsum[o * 3 + 2];
            SZ += s * (double)k;
          }
          rt_unit_finish(a, crop + sg, t0, row, kind, j, cb, n_cb, unitmax[row * segs + sg], S, SX, SY, SZ, runstat);
        }
      }
    }
    if (NP > 1) __syncthreads();  // (the next column block re-uses rowmax / rowsum)
   }
}

// Round 4: the decode epilogue, a WAVE per softmax unit.  The 64 lanes of a wave are the 64 columns of the
// column block (in a packed block: 2 or 4 crops' segments of 32 / 16 columns); the wave walks the unit's rows
// (1 for a 2D row, D for a joint's depth slices) twice -- maximum, then exp and the four sums per column in
// f64 -- and reduces across its lanes with DPP / two cross-row shuffles: no barrier, no row statistics
// through LDS, no serial "one thread adds eight rows" tail (rounds 1 - 3: a 16-lane group per ROW, three
// passes with two workgroup barriers between them; 3.5 of the 24.5 us of the 64-crop launch, and 20 us of the
// 72-bin launch whose 73-row units one 16-lane group summed).  Units are dealt to the decoding waves in the
// order of their first row (the loader wave does not decode).  The statistics of a (unit, column block) --
// what cb-split launches hand to head_rt_merge_kernel -- come from this code in every kernel variant, so
// every dispatch choice still gives the same bits.
template <int RT, int NP, int NG>
__device__ __forceinline__ void rt_decode_units(const RtArgs& a, float* Ls, int* info_s, double* runstat, int tid,
                                                bool idle_wave, int HW, int crop, int t0, int cb0, int n_cb,
                                                int segs) {
  constexpr int R = RT * 16, LP = NP * kRtLP, NW = NG / 4;
  if (idle_wave) return;
  int tid_d = tid;
  asm volatile("" : "+v"(tid_d));  // (nothing below is computed in front of the K loop and held through it)
  const int lane = tid_d & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid_d >> 6);
  // rows that start a unit: a 2D row, or slice 0 of a joint's depth rows
  unsigned long long starts[2] = {0ull, 0ull};
#pragma unroll
  for (int k = 0; k < (R + 63) / 64; ++k) {
    const int row = k * 64 + lane;
    const unsigned inf = row < R ? (unsigned)info_s[row] : 0u;
    const int kind = inf & 3, d = (inf >> 2) & 0x3fff;
    starts[k] = __ballot(kind == 1 || (kind == 2 && d == 0));
  }
  const int segw = 64 / segs, seg = lane / segw, cis = lane - seg * segw;  // column inside its crop's segment
  const bool first_of_seg = cis == 0 && crop + seg < a.B;
#pragma unroll 1
  for (int np = 0; np < NP; ++np) {
    const int cb = NP == 1 ? cb0 : cb0 + np;
    if (NP > 1 && cb >= n_cb) break;
    const float* Lb = Ls + np * kRtLP + lane;  // this lane's column of the block (row pitch LP)
    const int p = cb * 64 + cis;               // its position in the crop's map
    const bool valid = p < HW;
    const int ph = valid ? p / a.W : 0, pw = valid ? p - ph * a.W : 0;
    const double dw = (double)pw, dh = (double)ph;
    int u = 0;
#pragma unroll
    for (int k = 0; k < (R + 63) / 64; ++k) {
      unsigned long long msk = starts[k];
      while (msk) {
        const int row = k * 64 + __builtin_ctzll(msk);
        msk &= msk - 1;
        const bool mine = u % NW == wave;
        ++u;
        if (!mine) continue;  // (wave-uniform)
        const unsigned inf = (unsigned)info_s[row];
        const int kind = inf & 3, j = (int)(inf >> 16);
        const int n = kind == 2 ? a.D : 1;
        const float* x = Lb + row * LP;
        float m;
        double col, sz;
        // rows in register chunks: all LDS reads of a chunk are issued before the first is used (a loop of
        // read - wait - use per row was 2 us slower than the three-pass decode it replaces)
        auto chunk_max = [&](int r0, auto ch_tag, float (&xr)[decltype(ch_tag)::value]) {
          constexpr int CH = decltype(ch_tag)::value;
          float mm = -INFINITY;
#pragma unroll
          for (int r = 0; r < CH; ++r) {
            const int rr = r0 + r < n ? r0 + r : n - 1;  // (clamped: a valid LDS address)
            xr[r] = x[rr * LP];
          }
#pragma unroll
          for (int r = 0; r < CH; ++r) {
            if (r0 + r >= n) xr[r] = -INFINITY;
            mm = fmaxf(mm, xr[r]);
          }
          return mm;
        };
        auto chunk_sums = [&](int r0, auto ch_tag, const float (&xr)[decltype(ch_tag)::value], float nm, float mf,
                              double& c_a, double& c_b, double& z_a, double& z_b) {
          constexpr int CH = decltype(ch_tag)::value;
#pragma unroll
          for (int r = 0; r < CH; ++r) {
            // (a row behind the unit holds -inf: exp = 0; the f64 polynomial path is masked instead)
            const double e = MTR_RT_EXP32 ? (double)exp_shifted(xr[r], nm)
                                          : (r0 + r < n ? exp_neg64((double)xr[r] - (double)mf) : 0.0);
            if (r & 1) { c_b += e; z_b += e * (double)(r0 + r); }
            else { c_a += e; z_a += e * (double)(r0 + r); }
          }
        };
        auto seg_max = [&](float v) {
          if (!valid) v = -INFINITY;
          return segs == 1 ? group_max<64>(v) : segs == 2 ? group_max<32>(v) : group_max<16>(v);
        };
        double c_a = 0.0, c_b = 0.0, z_a = 0.0, z_b = 0.0;
        if (n == 1) {
          float xr[1];
          m = seg_max(chunk_max(0, std::integral_constant<int, 1>{}, xr));
          chunk_sums(0, std::integral_constant<int, 1>{}, xr, -m * kLog2e, m, c_a, c_b, z_a, z_b);
        } else if (n <= 8) {
          float xr[8];
          m = seg_max(chunk_max(0, std::integral_constant<int, 8>{}, xr));
          chunk_sums(0, std::integral_constant<int, 8>{}, xr, -m * kLog2e, m, c_a, c_b, z_a, z_b);
        } else if (n <= 16) {
          float xr[16];
          m = seg_max(chunk_max(0, std::integral_constant<int, 16>{}, xr));
          chunk_sums(0, std::integral_constant<int, 16>{}, xr, -m * kLog2e, m, c_a, c_b, z_a, z_b);
        } else {  // long units (72 depth slices): two passes over the rows, 16 at a time
          float mm = -INFINITY;
          for (int r0 = 0; r0 < n; r0 += 16) {
            float xr[16];
            mm = fmaxf(mm, chunk_max(r0, std::integral_constant<int, 16>{}, xr));
          }
          m = seg_max(mm);
          for (int r0 = 0; r0 < n; r0 += 16) {
            float xr[16];
            chunk_max(r0, std::integral_constant<int, 16>{}, xr);
            chunk_sums(r0, std::integral_constant<int, 16>{}, xr, -m * kLog2e, m, c_a, c_b, z_a, z_b);
          }
        }
        if (!valid) c_a = c_b = z_a = z_b = 0.0;
        col = c_a + c_b;
        sz = z_a + z_b;
        double S = col, SX = col * dw, SY = col * dh, SZ = sz;
        if (segs == 1) {
          S = group_sum<64>(S); SX = group_sum<64>(SX); SY = group_sum<64>(SY); SZ = group_sum<64>(SZ);
        } else if (segs == 2) {
          S = group_sum<32>(S); SX = group_sum<32>(SX); SY = group_sum<32>(SY); SZ = group_sum<32>(SZ);
        } else {
          S = group_sum<16>(S); SX = group_sum<16>(SX); SY = group_sum<16>(SY); SZ = group_sum<16>(SZ);
        }
        if (first_of_seg) rt_unit_finish(a, crop + seg, t0, row, kind, j, cb, n_cb, m, S, SX, SY, SZ, runstat);
      }
    }
  }
}

// ... and for units of <= 16 rows (every shipped configuration: D = 8) a 16-LANE GROUP per unit: lane l of the
// group holds columns 4 l .. 4 l + 3 of ALL the unit's rows in registers (one ds_read_b128 per row, issued
// back to back), so the maximum, the exps and the four f64 sums need one pass and only DPP reductions inside
// the group (a packed block's 2 / 4 crop segments are its 8- / 4-lane halves / quads): no barrier, no LDS
// round trip, no cross-row shuffle, and the <= 9 units of a block of three tiles run at once on the 16 groups
// of the four decoding waves.
template <int RT, int NP, int NG>
__device__ __forceinline__ void rt_decode_groups(const RtArgs& a, float* Ls, int* info_s, double* runstat, int tid,
                                                 bool idle_wave, int HW, int crop, int t0, int cb0, int n_cb,
                                                 int segs) {
  constexpr int R = RT * 16, LP = NP * kRtLP;
  if (idle_wave) return;
  int tid_d = tid;
  asm volatile("" : "+v"(tid_d));
  const int lane = tid_d & 63, grp = tid_d >> 4, l16 = tid_d & 15;
  unsigned long long starts[2] = {0ull, 0ull};
#pragma unroll
  for (int k = 0; k < (R + 63) / 64; ++k) {
    const int row = k * 64 + lane;
    const unsigned inf = row < R ? (unsigned)info_s[row] : 0u;
    const int kind = inf & 3, d = (inf >> 2) & 0x3fff;
    starts[k] = __ballot(kind == 1 || (kind == 2 && d == 0));
  }
  const int seg_lanes = 16 / segs, seg = l16 / seg_lanes, li = l16 - seg * seg_lanes;
  const bool finisher = li == 0 && crop + seg < a.B;
#pragma unroll 1
  for (int np = 0; np < NP; ++np) {
    const int cb = NP == 1 ? cb0 : cb0 + np;
    if (NP > 1 && cb >= n_cb) break;
    const float* Lb = Ls + np * kRtLP + l16 * 4;  // this lane's four columns (row pitch LP)
    const int pbase = cb * 64 + li * 4;
    bool ok[4];
    float fw[4], fh[4];  // (small integers, exact in f32; widened where they meet the f64 sums)
    // h = p / W without four ~30-instruction integer divisions per lane: floor((p + 0.5) * (1 / W)) -- the true
    // quotient of p + 0.5 is at least 0.5 / W away from an integer and the f32 product is within p / W * 2^-22
    // of it, i.e. exact for every p < 2^21 (maps are a few thousand positions)
    const float rcp_w = __frcp_rn((float)a.W);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = pbase + q;
      ok[q] = p < HW;
      const int h = !ok[q] ? 0 : HW <= 65536 ? (int)(((float)p + 0.5f) * rcp_w) : p / a.W;  // (checked exhaustively to 2^16)
      fh[q] = (float)h;
      fw[q] = (float)(ok[q] ? p - h * a.W : 0);
    }
    // Round t: group g takes the (t NG + g)-th unit of the block (in row order).  Every lane finds ITS unit's first
    // row in a wave-uniform walk over the start mask (a compare and a select per unit), then all groups of the
    // wave run ONE copy of the body -- a group's own branch would run the four groups of a wave one after the
    // other with 16 of 64 lanes each.  All units go through the CH-row body (rows behind a unit hold -inf).
    const int n_units = __builtin_popcountll(starts[0]) + __builtin_popcountll(starts[1]);
    for (int t = 0; t * NG < n_units; ++t) {
      const int target = t * NG + grp;
      int row = -1, cnt = 0;
#pragma unroll
      for (int k = 0; k < (R + 63) / 64; ++k) {
        unsigned long long msk = starts[k];
        while (msk) {
          const int rb = k * 64 + __builtin_ctzll(msk);
          msk &= msk - 1;
          row = cnt == target ? rb : row;
          ++cnt;
        }
      }
      if (row < 0) continue;
      const unsigned inf = (unsigned)info_s[row];
      const int kind = inf & 3, j = (int)(inf >> 16);
      const int n = kind == 2 ? a.D : 1;
      auto unit = [&](auto ch_tag) {
        constexpr int CH = decltype(ch_tag)::value;
        v4f x[CH];
#pragma unroll
        for (int r = 0; r < CH; ++r) x[r] = *reinterpret_cast<const v4f*>(Lb + (row + (r < n ? r : n - 1)) * LP);
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < CH; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (r >= n || !ok[q]) x[r][q] = -INFINITY;  // (rows behind the unit, columns behind the map: exp = 0)
            m = fmaxf(m, x[r][q]);
          }
        m = rt_seg_max(m, segs);
        const float nm = -m * kLog2e;
        double col[4] = {0.0, 0.0, 0.0, 0.0}, sz = 0.0, szb = 0.0;
#pragma unroll
        for (int r = 0; r < CH; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double e = MTR_RT_EXP32 ? (double)exp_shifted(x[r][q], nm)
                                          : (x[r][q] == -INFINITY ? 0.0 : exp_neg64((double)x[r][q] - (double)m));
            col[q] += e;
            if (q & 1) szb += e * (double)r; else sz += e * (double)r;
          }
        double S = (col[0] + col[1]) + (col[2] + col[3]);
        double SX = (col[0] * (double)fw[0] + col[1] * (double)fw[1]) + (col[2] * (double)fw[2] + col[3] * (double)fw[3]);
        double SY = (col[0] * (double)fh[0] + col[1] * (double)fh[1]) + (col[2] * (double)fh[2] + col[3] * (double)fh[3]);
        double SZ = sz + szb;
        S = rt_seg_sum(S, segs);
        SX = rt_seg_sum(SX, segs);
        SY = rt_seg_sum(SY, segs);
        SZ = rt_seg_sum(SZ, segs);
        if (finisher) rt_unit_finish(a, crop + seg, t0, row, kind, j, cb, n_cb, m, S, SX, SY, SZ, runstat);
      };
      if (a.D <= 8) unit(std::integral_constant<int, 8>{});   // (wave-uniform)
      else unit(std::integral_constant<int, 16>{});
    }
  }
}

// LEAN: the wave-per-unit form for every unit (16 row registers per lane instead of 64: head_rt16_kernel is
// built for five workgroups per CU, 102 registers)
template <int RT, int NP, int NG, bool LEAN = false>
__device__ __forceinline__ void rt_decode_blocks(const RtArgs& a, float* Ls, float* rowmax, float* unitmax,
                                                 int* info_s, double* rowsum, double* runstat, int tid,
                                                 bool idle_wave, int HW, int crop, int t0, int cb0, int n_cb,
                                                 int segs = 1) {
  if (MTR_RT_DECODE_PER_ROW)
    rt_decode_rows<RT, NP, NG>(a, Ls, rowmax, unitmax, info_s, rowsum, runstat, tid, idle_wave, HW, crop, t0, cb0, n_cb,
                               segs);
  else if (!LEAN && a.D <= 16)  // (wave-uniform)
    rt_decode_groups<RT, NP, NG>(a, Ls, info_s, runstat, tid, idle_wave, HW, crop, t0, cb0, n_cb, segs);
  else
    rt_decode_units<RT, NP, NG>(a, Ls, info_s, runstat, tid, idle_wave, HW, crop, t0, cb0, n_cb, segs);
}

// Column `col` (0 .. 63) of a workgroup's tile: in a packed last block (RtArgs::pack_g) segment s = col /
// (64 / pack) of it is the tail of crop + s -- the column inside that crop's block and the byte offset of
// that crop's features from the first crop's (a crop past the batch re-reads the last one's; its
// segment is never decoded).  pack = 0: the column itself.
struct RtPackedCol { int col; unsigned crop_off; };
__device__ __forceinline__ RtPackedCol rt_packed_col(const RtArgs& a, int crop, int pack, int col, int esize = 4) {
  if (pack == 0) return RtPackedCol{col, 0u};
  const int segw = 64 / pack, seg = col / segw;
  const int c = min(crop + seg, a.B - 1) - crop;
  return RtPackedCol{col - seg * segw, (unsigned)c * (unsigned)a.C * (unsigned)(a.H * a.W) * (unsigned)esize};
}

// The loader wave of head_rt_ld_kernel (see rt_block): ALL copies of every stage of one K loop, NBUF - 1
// stages ahead of the MFMA waves.  Per stage: wait until this wave's copies of stage s have landed
// (those of s + 1 .. s + NBUF - 2 stay in flight), meet the MFMA waves at barrier B_s -- they have
// then also left the slot of stage s - 1 --, and refill that slot with stage s + NBUF - 1.  Job j <
// 2 RT: KiB j of the block's packed weight tiles; the other 8: the 1-KiB chunks of the column
// block's 64 positions x 32 channels (layouts as in rt_block).  The sources advance by scalar adds.
__host__ __device__ constexpr int rt_ld_nbuf(int rtmax) { return rtmax <= 3 ? MTR_RT_LD_NBUF : 4; }
__host__ __device__ constexpr int rt_ld_la(int rtmax) { return rtmax <= 3 ? MTR_RT_LD_LA : 3; }

template <int RT, bool NHWC, int NBUF, int LA, int ESIZE = 4>
__device__ __forceinline__ void rt_loader_loop(const RtArgs& a, unsigned lds0, const char* fcrop, int t0,
                                               int cb, int n_stages, int lane, int HW, int crop = 0,
                                               int pack = 0) {
  constexpr int STAGE = rt_stage_bytes(RT, 1, NHWC);
  constexpr int JOBS = 2 * RT + 8;
  static_assert(LA >= 1 && LA <= NBUF - 1 && JOBS * (LA - 1) <= 63, "s_waitcnt vmcnt is a 6-bit count");
  // (the per-lane offsets below are derived HERE: without the empty asm the compiler computes them once
  //  in front of the column-block loop and keeps JOBS registers alive through the MFMA waves' path)
  asm volatile("" : "+v"(lane));
  const char* wsrc = uniform_ptr(a.wt + (size_t)t0 * 2048);
  const char* fsrc = uniform_ptr(fcrop);
  const unsigned wstride = (unsigned)a.n_tiles * 2048u;
  const unsigned fstride = NHWC ? 128u : 32u * (unsigned)HW * 4u;
  unsigned vo[JOBS];
#pragma unroll
  for (int j = 0; j < JOBS; ++j) {
    if (j < 2 * RT) {
      vo[j] = (unsigned)lane * 16u + (unsigned)j * 1024u;
    } else {
      const int jb = j - 2 * RT;
      if constexpr (NHWC) {
        const int pos = jb * 8 + (lane >> 3), slot = (lane & 7) ^ ((pos >> 1) & 7);
        const RtPackedCol pc = rt_packed_col(a, crop, pack, pos, ESIZE);
        const int P = cb * 64 + pc.col;
        vo[j] = pc.crop_off + (unsigned)(P < HW ? P : 0) * (unsigned)a.C * (unsigned)ESIZE + slot * 16;
      } else {
        const int ch = jb * 4 + (lane >> 4);
        const RtPackedCol pc = rt_packed_col(a, crop, pack, (lane & 15) * 4, ESIZE);
        const int p = cb * 64 + pc.col;
        vo[j] = pc.crop_off + (unsigned)ch * (unsigned)HW * 4u + (unsigned)(p < HW ? p : 0) * 4u;
      }
    }
  }
  auto issue = [&](int slot) {
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)slot * STAGE));
#pragma unroll
    for (int j = 0; j < JOBS; ++j) {
      if (j < 2 * RT)
        rt_dma16(wsrc, vo[j], base + j * 1024);
      else
        rt_dma16(fsrc, vo[j], base + RT * 2048 + (j - 2 * RT) * (NHWC ? 1024 : kRtChunkNCHW));
    }
    wsrc = uniform_ptr(wsrc + wstride);
    fsrc = uniform_ptr(fsrc + fstride);
  };
  int slot = 0;
  for (int p = 0; p < LA && p < n_stages; ++p) {
    issue(slot);
    slot = slot + 1 == NBUF ? 0 : slot + 1;
  }
  for (int s = 0; s < n_stages; ++s) {
    const int rem = n_stages - 1 - s;  // stages behind s; min(rem, LA - 1) of them are in flight
    if (rem >= LA - 1) {
      rt_wait_vmcnt<JOBS * (LA - 1)>();
    } else {
      bool waited = false;
      if constexpr (LA - 1 >= 3) {
        if (rem == 2) { rt_wait_vmcnt<JOBS * 2>(); waited = true; }
      }
      if constexpr (LA - 1 >= 2) {
        if (rem == 1) { rt_wait_vmcnt<JOBS>(); waited = true; }
      }
      if (!waited) rt_wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (s + LA < n_stages && !(MTR_RT_ABLATE & 4)) {
      issue(slot);
      slot = slot + 1 == NBUF ? 0 : slot + 1;
    }
  }
}

// KS = 2 (head_rt_ks_kernel, 512 threads): waves 4 .. 7 are a second K group -- same tiles, same
// positions, the ODD 32-channel stages through a ring of their own.  Two waves per SIMD for the
// launches that give every CU one workgroup: the matrix pipe works for one wave while the other
// sits at its barrier / fragment reads.  The sums stay those of the one-group kernel BIT FOR BIT:
// the odd group only runs the MFMA chains and hands every finished 32-channel chain to the even
// group through LDS (hb: two buffers by iteration parity, a third for the drain); the even group adds the chains to
// its f32 running sums in stage order c0, c1, c2, ... and carries them into f64 at the same
// stage boundaries (after c7, c15, ...) as rt_block<KS = 1> does.
//
// LD (head_rt_ld_kernel, 320 threads): a FIFTH wave is the loader -- it issues every
// global_load_lds of a stage, waits for its own copies with a counted vmcnt and meets the four
// MFMA waves at the stage barrier; those never execute a copy or a vmcnt wait: barrier, fragment
// reads, MFMAs.  A copy costs an MFMA wave 60 - 185 issue cycles beside its MFMAs
// (MI355X_MICROARCH.md, "LDS-DMA piece issue cost") against ~23 in a wave that does nothing else,
// and with one MFMA wave per SIMD nothing else fills those gaps.  Same chains, same sums, same bits.
// cb_first / cb_count: the column blocks this workgroup decodes (all of them, or ONE when the
// blocks of a map are dealt to different workgroups and merged by head_rt_merge_kernel).
template <int RT, int NP, int RTMAX, bool NHWC, int KS = 1, bool LD = false>
__device__ __forceinline__ void rt_block(const RtArgs& a, char* smem, int crop, int t0, int cb_first = 0,
                                         int cb_count = 1 << 30, int pack = 0) {
  static_assert(!LD || (KS == 1 && NP == 1), "the loader wave serves one K group of one column block");
  constexpr int STAGE = rt_stage_bytes(RT, NP, NHWC);
  constexpr int kRtNbuf = LD ? rt_ld_nbuf(RTMAX) : (KS == 2 ? MTR_RT_KS_NBUF : rt_nbuf(RTMAX, NP, NHWC));
  constexpr bool kPair = MTR_RT_PAIR && KS == 1 && kRtNbuf == 4 && !LD;
  constexpr int LA = kPair ? 2 : kRtNbuf - 1;   // stages in flight behind the one being consumed
  constexpr int NG = 16 * KS;                   // 16-lane groups of the workgroup (decode: one row each)
  constexpr int CHUNK = rt_feat_chunk(NHWC);   // LDS bytes of one column block's features per stage
  constexpr int JOBS = 2 * RT + 8 * NP;  // 1 KiB copies per stage: 2 per weight tile, 8 per column block
  constexpr int JPW = (JOBS + 3) / 4;    // per wave (upper bound)
  constexpr int R = RT * 16;
  constexpr int NA = RT * NP;            // accumulators per wave
  constexpr int LP = NP * kRtLP;         // logits row pitch (floats)
  float* Ls = reinterpret_cast<float*>(smem + KS * kRtNbuf * rt_stage_bytes(RTMAX, NP, NHWC));
  float* rowmax = Ls + RTMAX * 16 * LP;
  float* unitmax = rowmax + RTMAX * 16;
  float* bias_s = unitmax + RTMAX * 16;
  int* info_s = reinterpret_cast<int*>(bias_s + RTMAX * 16);
  double* rowsum = reinterpret_cast<double*>(info_s + RTMAX * 16);  // [R][3]
  double* runstat = rowsum + RTMAX * 16 * 3;                        // [R][5], maps of > 64 positions
  // KS = 2: finished chains of the odd group, [iteration parity][position group][accumulator][lane]
  v4f* hb = reinterpret_cast<v4f*>(smem + KS * kRtNbuf * rt_stage_bytes(RTMAX, NP, NHWC) +
                                   rt_epilogue_bytes(RTMAX, NP));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wid = wave & 3, kg = KS == 2 ? wave >> 2 : 0;  // position group, K group
  const bool is_loader = LD && wave == 4;                   // (wave-uniform)
  const int HW = a.H * a.W;
  const int n_stages = KS == 2 ? a.n_stages / 2 : a.n_stages;  // of this K group (KS = 2: C % 64 == 0)
  char* ring = smem + kg * kRtNbuf * STAGE;
  const int i16 = lane & 15, g4 = lane >> 4;
  const unsigned lds0 = rt_lds_addr(ring);
  const char* fcrop = reinterpret_cast<const char*>(a.feat) + (size_t)crop * a.C * HW * 4;
  const bool c_tail = KS == 1 && (a.C & 31) != 0;
  const int c0_last = (n_stages - 1) * 32;

  if (tid < R) {
    bias_s[tid] = a.bias_p[t0 * 16 + tid];
    info_s[tid] = a.info[t0 * 16 + tid];
  }

  // fragment addresses inside a stage buffer (bytes)
  const int a_off = i16 * 128 + ((g4 ^ ((i16 >> 1) & 7)) << 4);  // ^ 64 for the stage's second chain
  int b_off, b_q1;
  if constexpr (NHWC) {
    const int pos = wid * 16 + i16;
    b_off = pos * 128 + ((g4 ^ ((pos >> 1) & 7)) << 4);
    b_q1 = b_off ^ 64;
  } else {
    b_off = g4 * kRtChunkNCHW + (wid * 16 + i16) * 4;
    b_q1 = b_off + 4 * kRtChunkNCHW;
  }

  const int n_cb = (HW + 63) >> 6;
  const int cb_end = cb_first + cb_count < n_cb ? cb_first + cb_count : n_cb;
  for (int cb0 = cb_first; cb0 < cb_end; cb0 += NP) {  // groups of NP column blocks: one K loop each
    if constexpr (LD) {
      if (is_loader)
        rt_loader_loop<RT, NHWC, kRtNbuf, rt_ld_la(RTMAX)>(a, lds0, fcrop, t0, cb0, n_stages, lane, HW, crop, pack);
    }
    if (!is_loader) {
    // ---- this wave's copies: job j = wid + 4 i (0 .. 2RT-1: weight tiles, then 8 feature chunks).
    // A wave issues JPW copies per stage, or JPW - 1 when its last index falls behind the list
    // (wave-uniform; its counted s_waitcnt is one smaller per stage in flight).  Round 2: such a wave
    // used to repeat its previous copy for the sake of one wait count -- 2 of 16 KiB per stage at
    // 3 tiles, and the copies are what bounds the K loop (tools/experiments/ablate_rt.py, a27).
    const char* gbase[JPW];
    unsigned gstride[JPW], voff[JPW], ldso[JPW];
    // (the per-lane offsets are derived HERE, per column block: without the empty asm the compiler computes
    //  the parts that do not depend on the block -- a packed block's crop offsets -- once in front of the loop
    //  and holds them through the K loop AND the decode; in the 256-register instantiations that is a spill)
    int lane_j = lane;
    asm volatile("" : "+v"(lane_j));
#define lane lane_j
#pragma unroll
    for (int i = 0; i < JPW; ++i) {
      const int j = wid + 4 * i < JOBS ? wid + 4 * i : wid + 4 * (i - 1);
      if (j < 2 * RT) {
        gbase[i] = a.wt + (size_t)t0 * 2048 + j * 1024;
        gstride[i] = (unsigned)a.n_tiles * 2048u;
        voff[i] = lane * 16;
        ldso[i] = j * 1024;
      } else {
        const int jf = j - 2 * RT;  // column block of the group, 1-KiB chunk inside it
        const int np = NP == 1 ? 0 : jf >> 3, jb = NP == 1 ? jf : jf & 7;
        gbase[i] = fcrop;
        if constexpr (NHWC) {
          const int pos = jb * 8 + (lane >> 3), slot = (lane & 7) ^ ((pos >> 1) & 7);
          const RtPackedCol pc = rt_packed_col(a, crop, pack, pos);
          const int P = (cb0 + np) * 64 + pc.col;
          gstride[i] = 128;
          voff[i] = pc.crop_off + (unsigned)(P < HW ? P : 0) * (unsigned)a.C * 4u + slot * 16;
          ldso[i] = RT * 2048 + np * CHUNK + jb * 1024;
        } else {
          const int ch = jb * 4 + (lane >> 4);
          const RtPackedCol pc = rt_packed_col(a, crop, pack, (lane & 15) * 4);
          const int p = (cb0 + np) * 64 + pc.col;
          gstride[i] = 32u * (unsigned)HW * 4u;
          voff[i] = pc.crop_off + (unsigned)ch * (unsigned)HW * 4u + (unsigned)(p < HW ? p : 0) * 4u;
          ldso[i] = RT * 2048 + np * CHUNK + jb * kRtChunkNCHW;
        }
      }
      gbase[i] = uniform_ptr(gbase[i]);
      if constexpr (KS == 2) {  // K group g: stages g, g + 2, ...
        voff[i] += (unsigned)kg * gstride[i];
        gstride[i] *= 2;
      }
    }
#undef lane
    // copy i of the next stage into ring slot `slot` (stages are issued in order: the per-lane
    // offset walks along K; the last stage of a C that is not a multiple of 32 redirects the lanes
    // whose channels do not exist to ones that do -- their products meet zero weights)
    int issued = 0;
    // (a wave whose last index falls behind the job list has one copy less per stage: wave-uniform)
    const bool short_wave = JOBS % 4 != 0 && wid + 4 * (JPW - 1) >= JOBS;
    auto issue_job = [&](int i, int slot) {
      if (JOBS % 4 != 0 && i == JPW - 1 && short_wave) return;
      rt_dma16(gbase[i], voff[i], lds0 + (unsigned)slot * STAGE + ldso[i]);
      voff[i] += gstride[i];
    };
    auto redirect_tail = [&]() {  // (rare path: recomputed here rather than held in registers)
#pragma unroll
      for (int i = 0; i < JPW; ++i) {
        const int j = wid + 4 * i < JOBS ? wid + 4 * i : wid + 4 * (i - 1);
        if (j < 2 * RT) continue;
        const int jb = NP == 1 ? j - 2 * RT : (j - 2 * RT) & 7;
        if constexpr (NHWC) {
          const int pos = jb * 8 + (lane >> 3), slot = (lane & 7) ^ ((pos >> 1) & 7);
          if (c0_last + slot * 4 >= a.C) voff[i] -= (unsigned)(slot * 16);  // -> channel slot 0
        } else {
          const int ch = jb * 4 + (lane >> 4);
          if (c0_last + ch >= a.C) voff[i] -= (unsigned)ch * (unsigned)HW * 4u;  // -> channel row 0
        }
      }
    };
    auto stage_issued = [&]() {
      ++issued;
      if (c_tail && issued == n_stages - 1) redirect_tail();
    };

    RtRegs<RT, NP> rg;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rg.acc[q][r] = 0.0;
      rg.part[0][q] = rg.part[1][q] = rg.run[q] = v4f{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) rg.ya[t] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < NP; ++p) rg.yb[p] = v4f{0.f, 0.f, 0.f, 0.f};

    if constexpr (KS == 2) {  // the even group reads hb from its first iteration on: no chain yet = zeros
      if (kg == 1) {
#pragma unroll
        for (int q = 0; q < 2 * NA; ++q)
          hb[(q / NA * 4 + wid) * NA * 64 + (q % NA) * 64 + lane] = v4f{0.f, 0.f, 0.f, 0.f};
      }
    }
    // prologue: stages 0 .. NBUF-2 in flight
    if constexpr (!LD) {
      if (c_tail && n_stages == 1) redirect_tail();
#pragma unroll
      for (int p = 0; p < LA; ++p)
        if (p < n_stages) {
#pragma unroll
          for (int i = 0; i < JPW; ++i) issue_job(i, p);
          stage_issued();
        }
    }

    // Iteration of stage s with parity P = s & 1 and ring slot BUF = s % NBUF (literals: the chains
    // are registers, the LDS addresses immediates):
    //   wait for this wave's copies of stage s (those of s+1 .. s+NBUF-2 stay in flight), barrier
    //   (all copies of s visible; everyone finished reading the slot of s-1); read the first
    //   half's fragments of s; SECOND half of stage s-1's chain from the fragments read before the
    //   barrier (covers the LDS latency), with the copies of stage s+NBUF-1 into the slot of s-1
    //   in the MFMAs' shadow; read the second half's fragments; first half of stage s's chain with
    //   the finished chain of s-1 added to the f32 running sums underneath.  Every kRtCarry stages
    //   the running sums go into f64.
    // The wave issues in order: whatever sits between two MFMAs runs in the first one's 32-cycle
    // shadow (about 5 instructions), a longer run of scalar or vector work idles the matrix pipe --
    // so the loop is unrolled over the ring (no index arithmetic), MORE is a literal in the main
    // loop (no branches around the copies), and an MFMA is followed by at most one copy or one
    // packed add, pinned by a scheduling fence.
    // ROLE (literal): 0 = the only K group; 1 = even group of two: before its own chain of stage
    // 2S-2 (second block) it adds the odd group's chain of stage 2S-3 (first block; hb was zeroed
    // for the iterations that have none), and carries into f64 between the two when 2S-3 = 7 (mod 8);
    // 2 = odd group: no sums, every finished chain goes to hb[S & 1]
#define RT_BODY(S, P, BUF, MORE, ROLE)                                                            \
  {                                                                                               \
    const bool more = (MORE) && !LD;                                                              \
    if (LD) { /* the loader wave waited for the copies of stage S; nothing of ours is in flight */ \
      __syncthreads();                                                                            \
    } else if (kPair) { /* even stages: this wave's copies of stages S and S + 1 have landed, then meet */ \
      if (((S) & 1) == 0) { rt_wait_vmcnt<0>(); __syncthreads(); }                                \
    } else {                                                                                      \
      if (!more || kRtNbuf == 2) rt_wait_vmcnt<0>();                                              \
      else if (short_wave) rt_wait_vmcnt<(kRtNbuf - 2) * (JPW - 1)>();                            \
      else rt_wait_vmcnt<(kRtNbuf - 2) * JPW>();                                                  \
      __syncthreads();                                                                            \
    }                                                                                             \
    const char* buf = ring + (BUF) * STAGE;                                                       \
    v4f xa[RT], xb[NP];                                                                           \
    constexpr int R0 = LD ? 0 : JPW;   /* first MFMA slot of block A that takes a fragment read */ \
    if (!MTR_RT_SPREAD_READS) rt_read_frags<RT, NP, NHWC>(buf, a_off, b_off, xa, xb);             \
    v4f lr[NA];                                                                                   \
    if (ROLE == 1) {                                                                              \
      _Pragma("unroll") for (int q = 0; q < NA; ++q)                                              \
        lr[q] = hb[((((S) + 1) & 1) * 4 + wid) * NA * 64 + q * 64 + lane];                        \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    _Pragma("unroll") for (int n = 0; n < 4 * NA; ++n) {                                          \
      rt_mfma_slot<RT, NP>(rg.part[(P) ^ 1], rg.ya, rg.yb, n, false);                             \
      if (n < JPW && more && !(MTR_RT_ABLATE & 4)) issue_job(n, ((BUF) + LA) % kRtNbuf);          \
      if (n == JPW && more) stage_issued();                                                       \
      if (ROLE == 1 && n >= 2 * NA) rt_run_add<NA>(rg.run, lr, n - 2 * NA);                       \
      if (MTR_RT_SPREAD_READS && n >= R0 && n - R0 < RT + NP)                                     \
        rt_read_frag_unit<RT, NP, NHWC>(buf, a_off, b_off, xa, xb, n - R0);                       \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    if (MTR_RT_SPREAD_READS) {                                                                    \
      _Pragma("unroll") for (int u = (4 * NA - R0 > 0 ? 4 * NA - R0 : 0); u < RT + NP; ++u)       \
        rt_read_frag_unit<RT, NP, NHWC>(buf, a_off, b_off, xa, xb, u);                            \
    }                                                                                             \
    if (ROLE == 1 && (S) >= 5 && ((S) - 1) % 4 == 0) rt_flush<NA>(rg.acc, rg.run);                \
    if (!MTR_RT_SPREAD_READS) rt_read_frags<RT, NP, NHWC>(buf, a_off ^ 64, b_q1, rg.ya, rg.yb);   \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    _Pragma("unroll") for (int n = 0; n < 4 * NA; ++n) {                                          \
      rt_mfma_slot<RT, NP>(rg.part[P], xa, xb, n, true);                                          \
      if (MTR_RT_SPREAD_READS && n < RT + NP)                                                     \
        rt_read_frag_unit<RT, NP, NHWC>(buf, a_off ^ 64, b_q1, rg.ya, rg.yb, n);                  \
      if (ROLE != 2) {                                                                            \
        if (n % 2 == 1) rt_run_add<NA>(rg.run, rg.part[(P) ^ 1], n / 2);                          \
      } else if (n % 2 == 1 && n / 2 < NA) {                                                      \
        hb[(((S) & 1) * 4 + wid) * NA * 64 + (n / 2) * 64 + lane] = rg.part[(P) ^ 1][n / 2];      \
      }                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    if (MTR_RT_SPREAD_READS) {                                                                    \
      _Pragma("unroll") for (int u = 4 * NA; u < RT + NP; ++u)                                    \
        rt_read_frag_unit<RT, NP, NHWC>(buf, a_off ^ 64, b_q1, rg.ya, rg.yb, u);                  \
    }                                                                                             \
  }
#define RT_ITER(S, P, BUF, MORE)                                                                  \
  {                                                                                               \
    if constexpr (KS == 1) RT_BODY(S, P, BUF, MORE, 0)                                            \
    else if (kg == 0) RT_BODY(S, P, BUF, MORE, 1)                                                 \
    else RT_BODY(S, P, BUF, MORE, 2)                                                              \
  }
    static_assert(kRtNbuf == 8 || kRtNbuf == 4 || kRtNbuf == 2, "the main loop is unrolled over a ring of 8, 4 or 2 slots");
    static_assert(JPW < 4 * NA, "the copies fit the first half stage");
    int s = 0;
    // main loop: every iteration issues a stage.  run holds the stages up to s - 2 at the top of
    // iteration s; it is emptied into f64 every kRtCarry stages (any point between two iterations
    // is a valid one).
    if constexpr (kRtNbuf == 8) {
      for (; s + LA + 7 < n_stages; s += 8) {
        RT_ITER(s, 0, 0, true)
        if (KS == 1 && s >= kRtCarry && s % kRtCarry == 0) rt_flush<NA>(rg.acc, rg.run);
        RT_ITER(s + 1, 1, 1, true)
        RT_ITER(s + 2, 0, 2, true)
        RT_ITER(s + 3, 1, 3, true)
        RT_ITER(s + 4, 0, 4, true)
        RT_ITER(s + 5, 1, 5, true)
        RT_ITER(s + 6, 0, 6, true)
        RT_ITER(s + 7, 1, 7, true)
      }
    } else if constexpr (kRtNbuf == 4) {
      for (; s + LA + 3 < n_stages; s += 4) {
        RT_ITER(s, 0, 0, true)
        if (KS == 1 && s >= kRtCarry && s % kRtCarry == 0) rt_flush<NA>(rg.acc, rg.run);
        RT_ITER(s + 1, 1, 1, true)
        RT_ITER(s + 2, 0, 2, true)
        RT_ITER(s + 3, 1, 3, true)
      }
    } else {
      for (; s + LA + 1 < n_stages; s += 2) {
        RT_ITER(s, 0, 0, true)
        if (KS == 1 && s >= kRtCarry && s % kRtCarry == 0) rt_flush<NA>(rg.acc, rg.run);
        RT_ITER(s + 1, 1, 1, true)
      }
    }
    // remainder (the last issuing iterations + the NBUF - 1 that only consume)
    for (; s < n_stages; ++s) {
      if (KS == 1 && s >= 2 && (s - 1) % kRtCarry == 0) rt_flush<NA>(rg.acc, rg.run);
      const bool more_rt = s + LA < n_stages;
      if constexpr (kRtNbuf == 8) {
        switch (s & 7) {
          case 0: RT_ITER(s, 0, 0, more_rt) break;
          case 1: RT_ITER(s, 1, 1, more_rt) break;
          case 2: RT_ITER(s, 0, 2, more_rt) break;
          case 3: RT_ITER(s, 1, 3, more_rt) break;
          case 4: RT_ITER(s, 0, 4, more_rt) break;
          case 5: RT_ITER(s, 1, 5, more_rt) break;
          case 6: RT_ITER(s, 0, 6, more_rt) break;
          default: RT_ITER(s, 1, 7, more_rt) break;
        }
      } else if constexpr (kRtNbuf == 4) {
        switch (s & 3) {
          case 0: RT_ITER(s, 0, 0, more_rt) break;
          case 1: RT_ITER(s, 1, 1, more_rt) break;
          case 2: RT_ITER(s, 0, 2, more_rt) break;
          default: RT_ITER(s, 1, 3, more_rt) break;
        }
      } else {
        if (s & 1) RT_ITER(s, 1, 1, more_rt) else RT_ITER(s, 0, 0, more_rt)
      }
    }
#undef RT_ITER
#undef RT_BODY
    // drain: second half of the last stage's chain (parity PL), then the sums
#define RT_DRAIN(PL)                                                                              \
  {                                                                                               \
    _Pragma("unroll") for (int n = 0; n < 4 * NA; ++n) {                                          \
      rt_mfma_slot<RT, NP>(rg.part[PL], rg.ya, rg.yb, n, false);                                  \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    if constexpr (KS == 1) {                                                                      \
      _Pragma("unroll") for (int e2 = 0; e2 < 2 * NA; ++e2) rt_run_add<NA>(rg.run, rg.part[PL], e2); \
      if (MTR_RT_ABLATE & 8) {                                                                    \
        _Pragma("unroll") for (int t = 0; t < NA; ++t) rg.run[t] = rg.part[0][t] + rg.part[1][t]; \
      }                                                                                           \
      rt_flush<NA>(rg.acc, rg.run);                                                               \
    } else {                                                                                      \
      /* "iteration n_stages" of the hand-over: the odd group's last chain (stage 2 n_stages - 1  */ \
      /* of the K loop) goes to hb; the even group then adds, in stage order, the odd chain of    */ \
      /* stage 2 n_stages - 3, its own last chain and the odd group's last one                    */ \
      if (kg == 1) {                                                                              \
        _Pragma("unroll") for (int q = 0; q < NA; ++q)                                            \
          hb[(2 * 4 + wid) * NA * 64 + q * 64 + lane] = rg.part[PL][q];  /* a buffer of its own */ \
      }                                                                                           \
      __syncthreads();                                                                            \
      if (kg == 0) {                                                                              \
        if (n_stages >= 2) {                                                                      \
          v4f la[NA];                                                                             \
          _Pragma("unroll") for (int q = 0; q < NA; ++q)                                          \
            la[q] = hb[(((n_stages - 1) & 1) * 4 + wid) * NA * 64 + q * 64 + lane];               \
          _Pragma("unroll") for (int e2 = 0; e2 < 2 * NA; ++e2) rt_run_add<NA>(rg.run, la, e2);   \
          if (n_stages >= 5 && (n_stages - 1) % 4 == 0) rt_flush<NA>(rg.acc, rg.run);             \
        }                                                                                         \
        _Pragma("unroll") for (int e2 = 0; e2 < 2 * NA; ++e2) rt_run_add<NA>(rg.run, rg.part[PL], e2); \
        v4f lb[NA];                                                                               \
        _Pragma("unroll") for (int q = 0; q < NA; ++q)                                            \
          lb[q] = hb[(2 * 4 + wid) * NA * 64 + q * 64 + lane];                                    \
        _Pragma("unroll") for (int e2 = 0; e2 < 2 * NA; ++e2) rt_run_add<NA>(rg.run, lb, e2);     \
        rt_flush<NA>(rg.acc, rg.run);                                                             \
      }                                                                                           \
    }                                                                                             \
  }
    if ((n_stages - 1) & 1) RT_DRAIN(1) else RT_DRAIN(0)
#undef RT_DRAIN

    // ---- logits (+bias) -> LDS.  C/D layout of 16x16x4: col = l & 15, row = 4 (l >> 4) + reg
    if (kg == 0) {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));  // (as below: addresses computed here, not held through the K loop)
      const int col = wid * 16 + (lane_e & 15), row0 = (lane_e >> 4) * 4;
#pragma unroll
      for (int q = 0; q < NA; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (q % RT) * 16 + row0 + r;
          Ls[row * LP + (q / RT) * kRtLP + col] = (float)(rg.acc[q][r] + (double)bias_s[row]);
        }
    }
    }  // (!is_loader)
    __syncthreads();

    if (MTR_RT_ABLATE & 1) {  // no decode: one store per workgroup keeps the GEMM alive
      if (tid == 0 && cb0 + NP >= n_cb) a.c2d[(size_t)crop * a.J * 2] = Ls[0];
      continue;
    }
    if (pack == 0) {
      rt_decode_blocks<RT, NP, NG>(a, Ls, rowmax, unitmax, info_s, rowsum, runstat, tid, is_loader, HW, crop, t0,
                                   cb0, n_cb);
    } else {
      // a packed last block: per-row numbers once per crop segment -- 4x the arrays, in the ring (this
      // workgroup's only K loop is over: nothing is copied into it any more, every wave left it
      // before the barrier above)
      float* rowmax4 = reinterpret_cast<float*>(smem);  // (the first K group's ring: one set for the workgroup)
      float* unitmax4 = rowmax4 + R * 4;
      double* rowsum4 = reinterpret_cast<double*>(unitmax4 + R * 4);
      static_assert(R * 4 * (4 + 4 + 24) <= 2 * rt_stage_bytes(RT, NP, NHWC), "the segment statistics fit two ring slots");
      rt_decode_blocks<RT, NP, NG>(a, Ls, rowmax4, unitmax4, info_s, rowsum4, runstat, tid, is_loader, HW, crop, t0,
                                   cb0, n_cb, pack);
    }
    // (the next group's copies only touch the ring, which every wave left before the barrier behind
    //  the logits store; its logits store is many barriers away)
  }
}

// Block id -> (crop, block of tiles, column blocks).  XCD-aware (block id b runs on XCD b % 8): the
// workgroups of a crop -- its tile blocks and, when the column blocks of a map are dealt to
// different workgroups, those too -- share an XCD, so the crop's features come from HBM once and
// are re-read from that XCD's L2.
struct RtWork { int crop, blk, cb_first, cb_count, pack; };
__host__ __device__ inline int rt_wgs_per_8_crops(int n_blocks, int cb_split, int pack_g) {
  if (pack_g) return 8 * n_blocks * (cb_split - 1) + (8 / pack_g) * n_blocks;
  return 8 * n_blocks * (cb_split ? cb_split : 1);
}
__device__ __forceinline__ RtWork rt_work(const RtArgs& a) {
  const int chunk = rt_wgs_per_8_crops(a.n_blocks, a.cb_split, a.pack_g);
  const int id = blockIdx.x, in = id % chunk;
  RtWork w;
  w.pack = 0;
  const int full = a.pack_g ? 8 * a.n_blocks * (a.cb_split - 1) : chunk;  // workgroups of whole column blocks
  if (in < full) {
    w.crop = (id / chunk) * 8 + (in % 8);
    const int rest = in / 8;
    w.blk = rest % a.n_blocks;
    w.cb_first = a.cb_split ? rest / a.n_blocks : 0;
    w.cb_count = a.cb_split ? 1 : (1 << 30);
  } else {  // the packed last blocks of pack_g consecutive crops
    const int t = in - full, groups = 8 / a.pack_g;
    w.crop = (id / chunk) * 8 + (t % groups) * a.pack_g;
    w.blk = t / groups;
    w.cb_first = a.cb_split - 1;
    w.cb_count = 1;
    w.pack = a.pack_g;
  }
  return w;
}

template <int RTMAX, bool NHWC>
__global__ __launch_bounds__(256, RTMAX <= 3 ? 1 : 2) void head_rt_kernel(RtArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RtWork w = rt_work(a);
  if (w.crop >= a.B) return;
  const int t0 = w.blk * a.rtg;
  const int rt = min(a.rtg, a.n_tiles - t0);
  if (rt == 1) rt_block<1, 1, RTMAX, NHWC>(a, smem, w.crop, t0, w.cb_first, w.cb_count, w.pack);
  if (rt == 2) rt_block<2, 1, RTMAX, NHWC>(a, smem, w.crop, t0, w.cb_first, w.cb_count, w.pack);
  if (rt == 3) rt_block<3, 1, RTMAX, NHWC>(a, smem, w.crop, t0, w.cb_first, w.cb_count, w.pack);
  if constexpr (RTMAX >= 5) {
    if (rt == 4) rt_block<4, 1, RTMAX, NHWC>(a, smem, w.crop, t0, w.cb_first, w.cb_count, w.pack);
    if (rt == 5) rt_block<5, 1, RTMAX, NHWC>(a, smem, w.crop, t0, w.cb_first, w.cb_count, w.pack);
  }
}

// Two K groups per workgroup (see rt_block): blocks of <= 3 tiles, C a multiple of 64.
template <int RTMAX, bool NHWC>
__global__ __launch_bounds__(512, 1) void head_rt_ks_kernel(RtArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RtWork w = rt_work(a);
  if (w.crop >= a.B) return;
  const int t0 = w.blk * a.rtg;
  const int rt = min(a.rtg, a.n_tiles - t0);
  if (rt == 1) rt_block<1, 1, RTMAX, NHWC, 2>(a, smem, w.crop, t0, w.cb_first, w.cb_count, w.pack);
  if (rt == 2) rt_block<2, 1, RTMAX, NHWC, 2>(a, smem, w.crop, t0, w.cb_first, w.cb_count, w.pack);
  if (rt == 3) rt_block<3, 1, RTMAX, NHWC, 2>(a, smem, w.crop, t0, w.cb_first, w.cb_count, w.pack);
}
__host__ __device__ constexpr int rt_ks_lds_bytes(int rtmax, bool nhwc) {
  return 2 * MTR_RT_KS_NBUF * rt_stage_bytes(rtmax, 1, nhwc) + rt_epilogue_bytes(rtmax, 1) +
         3 * 4 * rtmax * 1024;  // + the chain hand-over buffers (two by iteration parity + the drain's)
}

// Four MFMA waves + a loader wave (see rt_block<..., LD = true>): one workgroup per CU, blocks of
// <= RTMAX tiles, C a multiple of 32.
template <int RTMAX, bool NHWC>
__global__ __launch_bounds__(320, 1) void head_rt_ld_kernel(RtArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RtWork w = rt_work(a);
  if (w.crop >= a.B) return;
  const int t0 = w.blk * a.rtg;
  const int rt = min(a.rtg, a.n_tiles - t0);
  if (rt == 1) rt_block<1, 1, RTMAX, NHWC, 1, true>(a, smem, w.crop, t0, w.cb_first, w.cb_count, w.pack);
  if (rt == 2) rt_block<2, 1, RTMAX, NHWC, 1, true>(a, smem, w.crop, t0, w.cb_first, w.cb_count, w.pack);
  if (rt == 3) rt_block<3, 1, RTMAX, NHWC, 1, true>(a, smem, w.crop, t0, w.cb_first, w.cb_count, w.pack);
  if constexpr (RTMAX >= 5) {
    if (rt == 4) rt_block<4, 1, RTMAX, NHWC, 1, true>(a, smem, w.crop, t0, w.cb_first, w.cb_count, w.pack);
    if (rt == 5) rt_block<5, 1, RTMAX, NHWC, 1, true>(a, smem, w.crop, t0, w.cb_first, w.cb_count, w.pack);
  }
}
__host__ __device__ constexpr int rt_ld_lds_bytes(int rtmax, bool nhwc) {
  return rt_ld_nbuf(rtmax) * rt_stage_bytes(rtmax, 1, nhwc) + rt_epilogue_bytes(rtmax, 1);
}

// Tiles of several column blocks (maps of more than 64 positions): RT row tiles x NP column blocks
// per workgroup, RT * NP <= 4.  Same grid mapping; a ragged last block (RT = 2, odd tile count)
// runs the one-tile body.
template <int RT, int NP, bool NHWC>
__global__ __launch_bounds__(256, 1) void head_rt_np_kernel(RtArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RtWork w = rt_work(a);
  if (w.crop >= a.B) return;
  const int t0 = w.blk * a.rtg;
  const int rt = min(a.rtg, a.n_tiles - t0);
  if (rt == RT) rt_block<RT, NP, RT, NHWC>(a, smem, w.crop, t0);
  if constexpr (RT == 2) {
    if (rt == 1) rt_block<1, NP, RT, NHWC>(a, smem, w.crop, t0);
  }
}

template <typename Kern>
static int rt_launch_kernel(Kern kern, int lds, const RtArgs& a, hipStream_t stream, int threads = 256) {
  if (lds > 64 * 1024) {
    const int rc = allow_dynamic_lds((const void*)kern, (size_t)lds);
    if (rc != MTR_OK) return rc;
  }
  const long long blocks = (long long)((a.B + 7) / 8) * rt_wgs_per_8_crops(a.n_blocks, a.cb_split, a.pack_g);
  if (blocks > 0x7fffffffLL) return MTR_E_SHAPE;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads), lds, stream, a);
  MTR_CHECK_LAUNCH();
  if (a.cb_split) {  // the column blocks' (max, sums) -> coordinates
    const long long rows = (long long)a.B * a.n_tiles * 16;
    hipLaunchKernelGGL(head_rt_merge_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, a,
                       a.cb_split);
    MTR_CHECK_LAUNCH();
  }
  return MTR_OK;
}

template <int RTMAX, bool NHWC>
static int rt_launch_t(const RtArgs& a, hipStream_t stream) {
  return rt_launch_kernel(head_rt_kernel<RTMAX, NHWC>, rt_lds_bytes(RTMAX, 1, NHWC), a, stream);
}
template <int RT, int NP, bool NHWC>
static int rt_launch_np(const RtArgs& a, hipStream_t stream) {
  return rt_launch_kernel(head_rt_np_kernel<RT, NP, NHWC>, rt_lds_bytes(RT, NP, NHWC), a, stream);
}
template <int RTMAX, bool NHWC>
static int rt_launch_ld(const RtArgs& a, hipStream_t stream) {
  return rt_launch_kernel(head_rt_ld_kernel<RTMAX, NHWC>, rt_ld_lds_bytes(RTMAX, NHWC), a, stream, 320);
}

size_t rt_workspace_bytes(int B, int J, int D, int H, int W) {
  if (!rt_shape_ok(1, J, D) || B <= 0 || H <= 0 || W <= 0) return 0;
  const int n_cb = (H * W + 63) / 64;
  if (n_cb < 2) return 0;
  return (size_t)B * n_cb * rt_geom(J, D).n_tiles * 16 * 5 * sizeof(double);
}

// =====================================================================================================
// 16-bit features (f16 / bf16), row-tile core: the shapes the joint-group kernels of head_fused.hip do
// not take -- more than 63 depth bins (1 + D > 64 rows per joint) or maps of more than 256 positions.
// Same row plan, same LDS images and the same loader wave as the f32 loader-wave kernel: a row of 64
// 16-bit channels is the 128 bytes a row of 32 f32 channels is, so a "stage" is 64 channels and the
// copy job list, the XOR swizzle and the fragment addresses are byte for byte those of the f32 NHWC
// path.  What differs is the arithmetic, and it is the reference's autocast arithmetic
// (multiperson_model.py:240-242): v_mfma_f32_16x16x32_{f16,bf16} on the features and on the weights
// ROUNDED TO THE FEATURE DTYPE, f32 accumulation along all of K in the matrix core (no f64 carry: the
// products of two 16-bit values are exact in f32), f32 logits on chip.  A lane's 16-byte fragment is 8
// consecutive channels = one operand of one MFMA: two MFMAs per row tile and stage instead of eight.
// Features must be NHWC (K-contiguous); NCHW features are transposed once into the caller's workspace
// (rt16_to_nhwc_kernel) -- these shapes spend 50 - 150 us in the head, the extra pass 3 - 10.
using h16x8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <typename T>
__device__ __forceinline__ v4f rt16_mfma(v4f a, v4f b, v4f c) {
  if constexpr (sizeof(T) == 2 && __is_same(T, __half))
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <typename T>
__global__ void head_rt16_pack_kernel(const float* __restrict__ w, const float* __restrict__ bias, int C,
                                      int J, int D, RtGeom g, int n_stages, char* __restrict__ section) {
  T* wt = reinterpret_cast<T*>(section);
  const size_t n_w = (size_t)n_stages * g.n_tiles * 1024;  // 16 rows x 64 channels per tile and stage
  float* bias_p = reinterpret_cast<float*>(section + n_w * 2);
  int* info = reinterpret_cast<int*>(bias_p + g.n_tiles * 16);
  const size_t total = n_w + (size_t)g.n_tiles * 16;
  for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total;
       u += (size_t)gridDim.x * blockDim.x) {
    if (u < n_w) {
      const int e = (int)(u & 7), slotp = (int)((u >> 3) & 7), row = (int)((u >> 6) & 15);
      const size_t ts = u >> 10;  // stage * n_tiles + tile
      const int tile = (int)(ts % g.n_tiles), stage = (int)(ts / g.n_tiles);
      const int c = stage * 64 + ((slotp ^ ((row >> 1) & 7)) << 3) + e;
      const RtRow rr = rt_row(g, J, D, tile * 16 + row);
      float v = 0.0f;
      if (rr.kind && c < C) v = w[(size_t)(rr.kind == 1 ? rr.joint : J + rr.d * J + rr.joint) * C + c];
      if constexpr (__is_same(T, __half)) wt[u] = __float2half(v);
      else wt[u] = __float2bfloat16(v);
    } else {
      const int r = (int)(u - n_w);
      const RtRow rr = rt_row(g, J, D, r);
      bias_p[r] = rr.kind ? bias[rr.kind == 1 ? rr.joint : J + rr.d * J + rr.joint] : 0.0f;
      info[r] = rt_encode(rr);
    }
  }
}

// [B][C][HW] -> [B][HW][C], 16-bit elements, 64 x 64 tiles through LDS (C % 64 == 0, HW % 4 == 0)
template <typename T>
__global__ __launch_bounds__(256) void rt16_to_nhwc_kernel(const T* __restrict__ in, T* __restrict__ out, int C,
                                                           int HW) {
  __shared__ T tile[64][66];
  const int b = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const T* src = in + ((size_t)b * C + c0) * HW;
  T* dst = out + ((size_t)b * HW) * C + c0;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ch = ty + 4 * i, p = p0 + tx;
    if (p < HW) tile[ch][tx] = src[(size_t)ch * HW + p];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int pl = ty + 4 * i, p = p0 + pl;
    if (p < HW) dst[(size_t)p * C + tx] = tile[tx][pl];
  }
}

// Occupancy instead of a deep ring: the MFMAs of a stage are ~1/16 of the f32 kernel's, so a workgroup
// is latency (barrier, LDS, copy), not matrix, bound.  Two ring slots, the loader one stage ahead, the
// decode's LDS aliased onto the ring (36 KiB per workgroup at 5 tiles) and <= 96 registers: FOUR
// workgroups = 20 waves per CU cover each other (first version: one workgroup per CU with a 4-slot
// ring, 102 KiB: B = 1024, D = 72: 901 us; library pair 455).
constexpr int kRt16Nbuf = 2, kRt16La = 1;
// [ring | decode scratch without the running statistics][running (max, 4 sums) per row across column blocks]
__host__ __device__ constexpr int rt16_scratch_bytes(int rtmax) {
  return kRt16Nbuf * rt_stage_bytes(rtmax, 1, true) > rt_epilogue_bytes(rtmax, 1) - rtmax * 16 * 40
             ? kRt16Nbuf * rt_stage_bytes(rtmax, 1, true) : rt_epilogue_bytes(rtmax, 1) - rtmax * 16 * 40;
}
__host__ __device__ constexpr int rt16_lds_bytes(int rtmax) { return rt16_scratch_bytes(rtmax) + rtmax * 16 * 40; }

template <typename T, int RT, int RTMAX>
__device__ __forceinline__ void rt16_block(const RtArgs& a, char* smem, int crop, int t0, int cb_first,
                                           int cb_count) {
  constexpr int STAGE = rt_stage_bytes(RT, 1, true);
  constexpr int R = RT * 16, LP = kRtLP;
  float* Ls = reinterpret_cast<float*>(smem);  // (the decode's arrays alias the ring: barriers around the K loop)
  float* rowmax = Ls + RTMAX * 16 * LP;
  float* unitmax = rowmax + RTMAX * 16;
  int* info_s = reinterpret_cast<int*>(unitmax + RTMAX * 16 * 2);
  double* rowsum = reinterpret_cast<double*>(info_s + RTMAX * 16);
  // (the running statistics of a map's column blocks outlive a K loop: behind the ring)
  double* runstat = reinterpret_cast<double*>(smem + rt16_scratch_bytes(RTMAX));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wid = wave & 3;
  const bool is_loader = wave == 4;
  const int HW = a.H * a.W;
  const int n_stages = a.n_stages;
  const unsigned lds0 = rt_lds_addr(smem);
  const char* fcrop = reinterpret_cast<const char*>(a.feat) + (size_t)crop * a.C * HW * 2;
  const int i16 = lane & 15, g4 = lane >> 4;
  const int a_off = i16 * 128 + ((g4 ^ ((i16 >> 1) & 7)) << 4);
  const int pos = wid * 16 + i16;
  const int b_off = RT * 2048 + pos * 128 + ((g4 ^ ((pos >> 1) & 7)) << 4);
  const int n_cb = (HW + 63) >> 6;
  const int cb_end = cb_first + cb_count < n_cb ? cb_first + cb_count : n_cb;
  for (int cb0 = cb_first; cb0 < cb_end; ++cb0) {
    if (is_loader) {
      rt_loader_loop<RT, true, kRt16Nbuf, kRt16La, 2>(a, lds0, fcrop, t0, cb0, n_stages, lane, HW);
    } else {
      v4f acc[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
      int slot = 0;
      for (int s = 0; s < n_stages; ++s) {
        __syncthreads();  // the loader waited for the copies of stage s; everyone left the slot of s - 1
        const char* buf = smem + slot * STAGE;
        slot = slot + 1 == kRt16Nbuf ? 0 : slot + 1;
        v4f fa0[RT], fa1[RT];
        const v4f fb0 = *reinterpret_cast<const v4f*>(buf + b_off);
        const v4f fb1 = *reinterpret_cast<const v4f*>(buf + (b_off ^ 64));
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          fa0[t] = *reinterpret_cast<const v4f*>(buf + t * 2048 + a_off);
          fa1[t] = *reinterpret_cast<const v4f*>(buf + t * 2048 + (a_off ^ 64));
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = rt16_mfma<T>(fa0[t], fb0, acc[t]);
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = rt16_mfma<T>(fa1[t], fb1, acc[t]);
      }
      __syncthreads();  // every wave has read its last fragments: the ring becomes the decode's scratch
      // logits (+bias) -> LDS.  C/D layout of the 16x16 accumulators: col = l & 15, row = 4 (l >> 4) + reg
      const int col = wid * 16 + i16, row0 = g4 * 4;
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = t * 16 + row0 + r;
          Ls[row * LP + col] = acc[t][r] + a.bias_p[t0 * 16 + row];
        }
      if (tid < R) info_s[tid] = a.info[t0 * 16 + tid];
    }
    if (is_loader) __syncthreads();  // (the loader's side of the barrier in front of the logits store)
    __syncthreads();
    rt_decode_blocks<RT, 1, 16, true>(a, Ls, rowmax, unitmax, info_s, rowsum, runstat, tid, is_loader, HW, crop, t0,
                                      cb0, n_cb);
    __syncthreads();  // the next column block's copies overwrite the decode's scratch
  }
}

template <typename T, int RTMAX>
__global__ __launch_bounds__(320, 5) void head_rt16_kernel(RtArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RtWork w = rt_work(a);
  if (w.crop >= a.B) return;
  const int t0 = w.blk * a.rtg;
  const int rt = min(a.rtg, a.n_tiles - t0);
  if (rt == 1) rt16_block<T, 1, RTMAX>(a, smem, w.crop, t0, w.cb_first, w.cb_count);
  if (rt == 2) rt16_block<T, 2, RTMAX>(a, smem, w.crop, t0, w.cb_first, w.cb_count);
  if (rt == 3) rt16_block<T, 3, RTMAX>(a, smem, w.crop, t0, w.cb_first, w.cb_count);
  if (rt == 4) rt16_block<T, 4, RTMAX>(a, smem, w.crop, t0, w.cb_first, w.cb_count);
  if (rt == 5) rt16_block<T, 5, RTMAX>(a, smem, w.crop, t0, w.cb_first, w.cb_count);
}

size_t rt16_section_bytes(int C, int J, int D) {
  if (!rt_shape_ok(C, J, D) || C % 64 != 0) return 0;
  const RtGeom g = rt_geom(J, D);
  return (size_t)(C / 64) * g.n_tiles * 2048 + (size_t)g.n_tiles * 16 * 8;
}

int rt16_pack(const float* weight, const float* bias, int C, int J, int D, int feat_dtype, void* section,
              hipStream_t stream) {
  const RtGeom g = rt_geom(J, D);
  const int n_stages = C / 64;
  const size_t total = (size_t)n_stages * g.n_tiles * 1024 + (size_t)g.n_tiles * 16;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  MTR_CLEAR_STALE();
  if (feat_dtype == MTR_F16)
    hipLaunchKernelGGL(head_rt16_pack_kernel<__half>, dim3((unsigned)blocks), dim3(256), 0, stream, weight, bias,
                       C, J, D, g, n_stages, (char*)section);
  else
    hipLaunchKernelGGL(head_rt16_pack_kernel<__hip_bfloat16>, dim3((unsigned)blocks), dim3(256), 0, stream, weight,
                       bias, C, J, D, g, n_stages, (char*)section);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

// workspace: [NHWC copy of NCHW features, 256-byte aligned size][column-block statistics]
static size_t rt16_nhwc_bytes(int B, int C, int H, int W) {
  return (((size_t)B * C * H * W * 2) + 255) & ~(size_t)255;
}
size_t rt16_workspace_bytes(int B, int C, int J, int D, int H, int W, int layout) {
  if (!rt_shape_ok(C, J, D) || C % 64 != 0 || B <= 0) return 0;
  return (layout == MTR_NCHW ? rt16_nhwc_bytes(B, C, H, W) : 0) + rt_workspace_bytes(B, J, D, H, W);
}

RtDispatch rt16_dispatch(int B, int H, int W, int J, int D, int rtg_hint, int split_hint, bool have_split_ws) {
  const RtGeom g = rt_geom(J, D);
  RtDispatch d{kRtKernel16, g.a > 1 ? g.a : 5, 1, 0, 0};
  const int n_cb = (H * W + 63) / 64;
  long long crops = (long long)((B + 7) / 8) * 8;
  // the MFMAs are 1/16 of the f32 kernel's: the launch is bounded by its copies, so the largest block
  // (fewest re-copies of a crop's features) that still gives every CU a workgroup; column blocks go to
  // different workgroups while the launch would otherwise leave CUs idle
  if (n_cb >= 2 && have_split_ws && split_hint != 1 &&
      (split_hint == 2 || crops * ((g.n_tiles + d.rtg - 1) / d.rtg) < 512)) {
    d.split = n_cb;
    crops *= n_cb;
  }
  if (g.a == 1) {
    int r = 5;
    while (r > 1 && crops * ((g.n_tiles + r - 1) / r) < 256) --r;
    if (rtg_hint >= 1 && rtg_hint <= 5) r = rtg_hint;
    d.rtg = r;
  }
  d.n_wg = crops * ((g.n_tiles + d.rtg - 1) / d.rtg);
  return d;
}

int rt16_launch(const void* feat, int feat_dtype, int layout, const void* section, int B, int C, int H, int W,
                int J, int D, const HeadScale& hs, float* coords2d, float* coords3d_rel, int rtg_hint,
                int split_hint, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  const RtGeom g = rt_geom(J, D);
  RtArgs a;
  a.n_stages = C / 64;
  a.wt = (const char*)section;
  a.bias_p = (const float*)(a.wt + (size_t)a.n_stages * g.n_tiles * 2048);
  a.info = (const int*)(a.bias_p + g.n_tiles * 16);
  a.B = B; a.C = C; a.H = H; a.W = W; a.J = J; a.D = D;
  a.n_tiles = g.n_tiles;
  a.hs = hs;
  a.inv = make_axis_inv(W, H, D);
  a.c2d = coords2d;
  a.c3d = coords3d_rel;
  char* ws = (char*)workspace;
  size_t left = workspace ? workspace_bytes : 0;
  const void* nhwc = feat;
  if (layout == MTR_NCHW) {  // K-contiguous operands: one transposing pass into the workspace
    const size_t need = rt16_nhwc_bytes(B, C, H, W);
    if (left < need) return MTR_E_WORKSPACE;
    const dim3 grid((unsigned)((H * W + 63) / 64), (unsigned)(C / 64), (unsigned)B);
    MTR_CLEAR_STALE();
    if (feat_dtype == MTR_F16)
      hipLaunchKernelGGL(rt16_to_nhwc_kernel<__half>, grid, dim3(256), 0, stream, (const __half*)feat, (__half*)ws, C,
                         H * W);
    else
      hipLaunchKernelGGL(rt16_to_nhwc_kernel<__hip_bfloat16>, grid, dim3(256), 0, stream,
                         (const __hip_bfloat16*)feat, (__hip_bfloat16*)ws, C, H * W);
    MTR_CHECK_LAUNCH();
    nhwc = ws;
    ws += need;
    left -= need;
  }
  a.feat = (const float*)nhwc;
  const bool can_split = left >= rt_workspace_bytes(B, J, D, H, W) && (H * W + 63) / 64 >= 2;
  const RtDispatch d = rt16_dispatch(B, H, W, J, D, rtg_hint, split_hint, can_split);
  a.cb_split = d.split;
  a.pack_g = 0;
  a.ws = d.split ? (double*)ws : nullptr;
  a.rtg = d.rtg;
  a.n_blocks = (g.n_tiles + a.rtg - 1) / a.rtg;
  if (feat_dtype == MTR_F16)
    return rt_launch_kernel(head_rt16_kernel<__half, 5>, rt16_lds_bytes(5), a, stream, 320);
  return rt_launch_kernel(head_rt16_kernel<__hip_bfloat16, 5>, rt16_lds_bytes(5), a, stream, 320);
}

int rt_pack(const float* weight, const float* bias, int C, int J, int D, void* section,
            hipStream_t stream) {
  const RtGeom g = rt_geom(J, D);
  const int n_stages = (C + 31) / 32;
  const size_t total = (size_t)n_stages * g.n_tiles * 512 + (size_t)g.n_tiles * 16;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(head_rt_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, weight, bias, C,
                     J, D, g, n_stages, (char*)section);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

// ---- launch plan.  Measured model of a launch (tools/experiments/head_sweep.py, profiles/r03*_head_sweep.jsonl;
// C = 1280): a workgroup of r tiles alone on its CU takes about 5.5 + 6.3 r us (one K loop of 40
// stages; the loader-wave kernel 4.7 + 6.1 r); of two 256-thread workgroups sharing a CU the K loops
// take 1.75 - 2.4x as long (kRtPairedKLoop; the 5.5 us outside them do not); workgroups start in block order as slots free up, one per CU for the
// loader-wave kernel (320 threads, its ring and registers fill the CU), two for the others.  A
// crop's blocks are r, r, ..., (the rest) tiles, so the launch is simulated: an event queue over the
// CUs (a CU's next completion; only its own completion changes its residents), O(workgroups x log
// CUs), the result cached per shape by rt_dispatch.  The plan with the smallest estimate is taken.
// What the closed form of the first version (every block r tiles, whole rounds) missed: 320 crops,
// r = 5 is 640 equal workgroups on 512 slots = 1.25 rounds (98 us), r = 4 is 960 workgroups of 4, 4, 2
// tiles that pack the second round (92 us; the library pair: 96).
struct RtPlan { int mode; int rtg; double us; };  // mode 0 = head_rt_kernel, 1 = loader-wave kernel
constexpr int kRtModelCusDefault = 256;  // MI355X, SPX mode; no device (host-only mtr_head_plan in a CPU process)
// CUs of the calling thread's current device (what the launch will run on): the plan simulates the launch on
// that many -- an MI355X partition in CPX mode has 32 -- and its memo is keyed by it.  Cached per device id.
static int rt_device_cus() {
  static std::mutex mu;
  static int cached[64] = {0};
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) {
    (void)hipGetLastError();
    return kRtModelCusDefault;
  }
  std::lock_guard<std::mutex> lock(mu);
  if (dev < 64 && cached[dev] > 0) return cached[dev];
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
    (void)hipGetLastError();
    return kRtModelCusDefault;
  }
  if (dev < 64) cached[dev] = cus;
  return cus;
}
// two workgroups on a CU: how much longer their K loops run (the rest of a workgroup is unchanged), by
// tiles per block of the launch -- measured at 1024 crops (10,240 .. 2,048 workgroups): blocks of 2 and of
// 5 tiles share a CU well, blocks of 1 and 3 tiles are slower together than one after the other
constexpr double kRtPairedKLoop[6] = {0.0, 2.4, 1.8, 2.43, 1.8, 1.75};
struct RtWgTime { double fixed, loop; };  // a workgroup alone on its CU: fixed + loop us
static RtWgTime rt_wg_us(int mode, int r, int tiles, int k_loops, double stage_scale) {
  // a block of `tiles` row tiles in the kernel instantiated for blocks of r
  const double fixed = mode == 1 ? 4.7 : 5.5;
  const double per_tile = (mode == 1 ? 6.1 : (r <= 3 ? 6.3 : r == 4 ? 6.1 : 6.5)) * stage_scale;  // (measured alone: 30 us at r = 4)
  return RtWgTime{fixed + (k_loops - 1) * 3.0 /* a decode per further K loop */, k_loops * per_tile * tiles};
}
static double rt_launch_us(int mode, int r, long long crops, int n_tiles, int k_loops, double stage_scale,
                           int kRtModelCus) {
  const int per_crop = (n_tiles + r - 1) / r, last = n_tiles - (per_crop - 1) * r;
  const long long n_wg = crops * per_crop;
  const RtWgTime w_full = rt_wg_us(mode, r, r, k_loops, stage_scale), w_last = rt_wg_us(mode, r, last, k_loops, stage_scale);
  const int slots = mode == 1 ? 1 : 2;
  // (solo time, paired slowdown) of the two block sizes
  const double t_full = w_full.fixed + w_full.loop, t_last = w_last.fixed + w_last.loop;
  // (round 6) blocks of 2 tiles that pair up in a launch of at most ONE round of pairs are slower together than the
  // steady-state figure: configs[2]'s own launch (32 crops of 12x12, column blocks split: 360 workgroups, 104 CUs
  // paired) measured 32.8 us for a predicted 28.2 -- 2.2 x -- and the plan took it over the loader-wave kernel's
  // 216 solo workgroups of 4, 4, 2 tiles (predicted 29.1, measured 29.9; profiles/r06z_head_sweep.jsonl)
  const double paired = (r == 2 && n_wg <= 2LL * kRtModelCus) ? 2.2 : kRtPairedKLoop[r];
  const double f_full = (w_full.fixed + paired * w_full.loop) / t_full;
  const double f_last = (w_last.fixed + paired * w_last.loop) / t_last;
  if (n_wg > 32768) {  // far into the steady state: work over throughput
    const double work = (double)crops * ((per_crop - 1) * t_full * (slots == 2 ? f_full / 2 : 1.0) +
                                         t_last * (slots == 2 ? f_last / 2 : 1.0));
    return work / kRtModelCus;
  }
  // A CU's residents run at their solo speed (alone) or 1 / their paired slowdown (two).  At equal times
  // completions come before hand-outs and the emptier CU is served first, and a CU takes ONE waiting
  // workgroup per event (the dispatcher spreads a second round over the CUs that are free rather than
  // doubling up on the first one) -- a CU with a second free slot queues again.
  struct Cu { double rem[2], fac[2]; int n; double at; int ver; };
  std::vector<Cu> cu(kRtModelCus, Cu{{0.0, 0.0}, {1.0, 1.0}, 0, 0.0, 0});
  using Ev = std::tuple<double, int, int, int, int>;  // (time, 0 = completion / 1 = take one more, residents, CU, version)
  std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> events;
  long long next = 0;
  auto is_last = [&](long long w) { return (int)(w % per_crop) == per_crop - 1; };
  auto speed = [](const Cu& u, int i) { return u.n <= 1 ? 1.0 : 1.0 / u.fac[i]; };
  auto advance = [&](Cu& u, double now) {  // progress since the CU's last event; finished residents leave
    const double dt = now - u.at;
    int keep = 0;
    for (int i = 0; i < u.n; ++i) {
      const double rem = u.rem[i] - dt * speed(u, i);
      if (rem > 1e-9) { u.rem[keep] = rem; u.fac[keep] = u.fac[i]; ++keep; }
    }
    u.n = keep;
    u.at = now;
  };
  auto requeue = [&](int c, double now) {
    Cu& u = cu[c];
    ++u.ver;
    if (u.n < slots && next < n_wg) events.push(Ev(now, 1, u.n, c, u.ver));  // a free slot: take one more
    else if (u.n > 0) {
      double first = u.rem[0] / speed(u, 0);
      if (u.n == 2 && u.rem[1] / speed(u, 1) < first) first = u.rem[1] / speed(u, 1);
      events.push(Ev(now + first, 0, u.n, c, u.ver));
    }
  };
  for (int c = 0; c < kRtModelCus; ++c) requeue(c, 0.0);
  double now = 0.0;
  while (!events.empty()) {
    const Ev e = events.top();
    events.pop();
    const int c = std::get<3>(e);
    Cu& u = cu[c];
    if (std::get<4>(e) != u.ver) continue;
    now = std::get<0>(e);
    advance(u, now);
    if (std::get<1>(e) == 1 && u.n < slots && next < n_wg) {
      const bool l = is_last(next++);
      u.rem[u.n] = l ? t_last : t_full;
      u.fac[u.n] = l ? f_last : f_full;
      ++u.n;
    }
    requeue(c, now);
  }
  return now;
}
static RtPlan rt_plan(long long crops, const RtGeom& g, int k_loops, int C, int rtg_hint, int ld_hint, int cus) {
  const double stage_scale = ((C + 31) / 32) / 40.0;
  const bool ld_ok = C % 32 == 0 && ld_hint != 1;
  RtPlan best{0, g.a > 1 ? g.a : 3, 1e30};
  for (int mode = 0; mode <= (ld_ok ? 1 : 0); ++mode) {
    if (ld_hint == 2 && ld_ok && mode == 0) continue;
    for (int r = 1; r <= 5; ++r) {
      if (g.a > 1 && r != g.a) continue;                 // atoms of several tiles are one block
      if (g.a == 1 && rtg_hint >= 1 && rtg_hint <= 5 && r != rtg_hint) continue;
      const double us = rt_launch_us(mode, r, crops, g.n_tiles, k_loops, stage_scale, cus);
      if (us < best.us) best = RtPlan{mode, r, us};
    }
  }
  return best;
}

// Which kernel a launch takes (shared by rt_launch and the host-only mtr_head_plan)
static RtDispatch rt_dispatch_uncached(int B, int C, int H, int W, int J, int D, int rtg_hint, int np_hint,
                                       int ks_hint, int ld_hint, int split_hint, bool have_workspace, int cus);
RtDispatch rt_dispatch(int B, int C, int H, int W, int J, int D, int rtg_hint, int np_hint, int ks_hint,
                       int ld_hint, int split_hint, bool have_workspace) {
  // The plan simulates the launch (0.5 - 20 ms of host time): memoised per process -- a pure function of
  // the key, rebuilt identically on a miss.  The key holds the device's CU count and the batch size
  // ROUNDED UP TO A MULTIPLE OF 8, which is all the plan reads of it (crops are dealt to XCDs in eights):
  // a server whose box count changes from call to call meets an eighth of the keys.
  static std::mutex mu;
  static std::map<std::array<int, 13>, RtDispatch> cache;
  const int cus = rt_device_cus();
  const int B8 = (B + 7) / 8 * 8;
  const std::array<int, 13> key{B8, C, H, W, J, D, rtg_hint, np_hint, ks_hint, ld_hint, split_hint,
                                (int)have_workspace, cus};
  {
    std::lock_guard<std::mutex> lock(mu);
    const auto it = cache.find(key);
    if (it != cache.end()) return it->second;
  }
  const RtDispatch d = rt_dispatch_uncached(B8, C, H, W, J, D, rtg_hint, np_hint, ks_hint, ld_hint, split_hint,
                                            have_workspace, cus);
  std::lock_guard<std::mutex> lock(mu);
  if (cache.size() > 4096) cache.clear();
  cache[key] = d;
  return d;
}
static RtDispatch rt_dispatch_uncached(int B, int C, int H, int W, int J, int D, int rtg_hint, int np_hint,
                                       int ks_hint, int ld_hint, int split_hint, bool have_workspace, int cus) {
  const RtGeom g = rt_geom(J, D);
  RtDispatch d{kRtKernelPlain, 3, 1, 0, 0, 0.0, 0};
  const int n_cb = (H * W + 63) / 64;
  const long long crops8 = (long long)((B + 7) / 8) * 8;
  // ---- maps of more than 64 positions: deal the 64-position column blocks to DIFFERENT workgroups
  // (their (max, sums) per softmax unit go through `workspace` to head_rt_merge_kernel, ~3 us for the
  // second launch) instead of one workgroup running n_cb K loops back to back: a launch of B crops
  // then has the shape of a launch of B * n_cb crops of 64 positions.  split_hint: 0 = when the
  // model says it pays, 1 = never, 2 = whenever a workspace is there.
  RtPlan plan = rt_plan(crops8, g, n_cb, C, rtg_hint, ld_hint, cus);
  if (n_cb >= 2 && have_workspace && split_hint != 1) {
    // a last column block of 16 (32) positions -- 12x12, 20x20, 28x28 (8x12, 16x10) maps: those of 4 (2)
    // consecutive crops in one workgroup
    const int tail = (H * W) % 64;
    const int pack = tail == 16 ? 4 : tail == 32 ? 2 : 0;
    const long long units = pack ? crops8 * (n_cb - 1) + crops8 / pack : crops8 * n_cb;  // "crops" of 64 positions
    RtPlan sp = rt_plan(units, g, 1, C, rtg_hint, ld_hint, cus);
    sp.us += 3.0;  // the merge launch
    if (split_hint == 2 || sp.us < plan.us) {
      plan = sp;
      d.split = n_cb;
      d.pack = pack;
    }
  }
  // ---- maps of several column blocks, one-tile atoms, no split: RT x NP tiles (one K loop for NP column
  // blocks) -- a workgroup of one row tile that runs n_cb K loops of 8 MFMAs per stage back to back
  // is bound by its per-stage overhead, not by the matrix pipe.  Such a tile needs 70 - 160 KB of
  // LDS, i.e. one workgroup per CU, so it only pays while the launch is ONE round of workgroups
  // (measured, tools/experiments/head_rt_ab.py: B=16 24x24, 160 workgroups: 84 us against 103;
  // B=32 12x12, 320 workgroups = two rounds on a quarter of the CUs: 57 against 54; B=64 16x16,
  // 640: 115 against 97).  np_hint: 1 = never, 2..4 = that many column blocks, 0 = this rule.
  if (g.a == 1 && n_cb >= 2 && np_hint != 1 && !d.split && ld_hint != 2) {
    int np = np_hint;
    if (np == 0) {
      np = n_cb == 2 ? 2 : (n_cb == 3 ? 3 : 4);
      if (n_cb > 4 && (n_cb + 2) / 3 * 3 - n_cb < (n_cb + 3) / 4 * 4 - n_cb) np = 3;  // less padding
      if (crops8 * g.n_tiles > cus) np = 1;  // more than one round: two workgroups per CU (below)
    }
    if (np >= 2) {
      d.kernel = kRtKernelNp;
      d.np = np > 4 ? 4 : np;
      d.rtg = (d.np == 2 && rtg_hint == 2) ? 2 : 1;
      d.n_wg = crops8 * ((g.n_tiles + d.rtg - 1) / d.rtg);
      return d;
    }
  }
  d.rtg = rt_block_tiles(g, plan.rtg);
  d.n_wg = crops8 / 8 * rt_wgs_per_8_crops((g.n_tiles + d.rtg - 1) / d.rtg, d.split, d.pack);
  d.model_us = plan.us;
  // ---- four MFMA waves + a loader wave (320 threads, one workgroup per CU; same bits): the MFMA
  // waves issue no copies and no vmcnt waits.  ld_hint: 0 = the plan, 1 = never, 2 = whenever the
  // kernel can (C % 32 == 0).
  if (plan.mode == 1) {
    d.kernel = kRtKernelLoader;
    return d;
  }
  // two K groups per workgroup (512 threads; same bits as one group, see rt_block): only on request
  // since the loader-wave kernel exists (round 3: it is as fast or faster on every launch the two K
  // groups were for -- B = 64: 23.6 against 24.6 us).  ks_hint: 2 = whenever the kernel can (C % 64
  // == 0, blocks of <= 3 tiles); 0, 1 = one K group.
  if (C % 64 == 0 && d.rtg <= 3 && ks_hint == 2) d.kernel = kRtKernelTwoKGroups;
  return d;
}

int rt_launch(const float* feat, int layout, const void* section, int B, int C, int H, int W, int J,
              int D, const HeadScale& hs, float* coords2d, float* coords3d_rel, int rtg_hint, int np_hint,
              int ks_hint, int ld_hint, int split_hint, void* workspace, size_t workspace_bytes,
              hipStream_t stream) {
  const RtGeom g = rt_geom(J, D);
  RtArgs a;
  a.feat = feat;
  a.n_stages = (C + 31) / 32;
  a.wt = (const char*)section;
  a.bias_p = (const float*)(a.wt + (size_t)a.n_stages * g.n_tiles * 2048);
  a.info = (const int*)(a.bias_p + g.n_tiles * 16);
  a.B = B; a.C = C; a.H = H; a.W = W; a.J = J; a.D = D;
  a.n_tiles = g.n_tiles;
  a.hs = hs;
  a.inv = make_axis_inv(W, H, D);
  a.c2d = coords2d;
  a.c3d = coords3d_rel;
  const bool nhwc = layout == MTR_NHWC;
  const bool can_split = workspace != nullptr && ((uintptr_t)workspace % 8) == 0 &&
                         workspace_bytes >= rt_workspace_bytes(B, J, D, H, W);
  const RtDispatch d = rt_dispatch(B, C, H, W, J, D, rtg_hint, np_hint, ks_hint, ld_hint, split_hint, can_split);
  a.cb_split = d.split;
  a.pack_g = d.pack;
  a.ws = d.split ? (double*)workspace : nullptr;
  a.rtg = d.rtg;
  a.n_blocks = (g.n_tiles + a.rtg - 1) / a.rtg;
  switch (d.kernel) {
    case kRtKernelNp:
      if (a.rtg == 2) return nhwc ? rt_launch_np<2, 2, true>(a, stream) : rt_launch_np<2, 2, false>(a, stream);
      switch (d.np) {
        case 2: return nhwc ? rt_launch_np<1, 2, true>(a, stream) : rt_launch_np<1, 2, false>(a, stream);
        case 3: return nhwc ? rt_launch_np<1, 3, true>(a, stream) : rt_launch_np<1, 3, false>(a, stream);
        default: return nhwc ? rt_launch_np<1, 4, true>(a, stream) : rt_launch_np<1, 4, false>(a, stream);
      }
    case kRtKernelLoader:
      if (a.rtg <= 3) return nhwc ? rt_launch_ld<3, true>(a, stream) : rt_launch_ld<3, false>(a, stream);
      return nhwc ? rt_launch_ld<5, true>(a, stream) : rt_launch_ld<5, false>(a, stream);
    case kRtKernelTwoKGroups:
      return nhwc ? rt_launch_kernel(head_rt_ks_kernel<3, true>, rt_ks_lds_bytes(3, true), a, stream, 512)
                  : rt_launch_kernel(head_rt_ks_kernel<3, false>, rt_ks_lds_bytes(3, false), a, stream, 512);
    default:
      if (a.rtg <= 3) return nhwc ? rt_launch_t<3, true>(a, stream) : rt_launch_t<3, false>(a, stream);
      return nhwc ? rt_launch_t<5, true>(a, stream) : rt_launch_t<5, false>(a, stream);
  }
}

}  // namespace mtr
