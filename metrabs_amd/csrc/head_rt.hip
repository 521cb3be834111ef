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
#include "head_rt.h"

namespace mtr {

using v4f = __attribute__((ext_vector_type(4))) float;

constexpr int kRtNbuf = 4;        // LDS ring: 3 stages in flight behind the one being consumed
constexpr int kRtLP = 68;         // logits row pitch in LDS (floats)
constexpr int kRtChunkNCHW = 1088;  // 4 channel rows of 64 positions + 64 B: the two channel groups
                                    // a 32-lane ds_read_b32 group touches land 16 banks apart

__host__ __device__ constexpr int rt_stage_bytes(int rt, bool nhwc) {
  return rt * 2048 + (nhwc ? 8192 : 8 * kRtChunkNCHW);
}
__host__ __device__ constexpr int rt_lds_bytes(int rtmax, bool nhwc) {
  // ring + logits [R][68] + per row: max, unit max, bias (f32), label (i32), 3 f64 sums
  return kRtNbuf * rt_stage_bytes(rtmax, nhwc) + rtmax * 16 * (kRtLP * 4 + 16 + 24);
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
               : "s"(lds_addr), "v"(voff), "s"(sbase)
               : "memory");
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
  float* c2d;
  float* c3d;
};

template <int RT>
struct RtRegs {
  double acc[RT][4];
  v4f part[2][2][RT];  // [stage parity][16-channel chain of the stage][tile]
  v4f ya[RT], yb;      // fragments of the previous stage's second chain (consumed one barrier late)
};

template <int RT, bool NHWC>
__device__ __forceinline__ void rt_read_frags(const char* buf, int a_addr, int b_addr, v4f (&fa)[RT],
                                              v4f& fb) {
#pragma unroll
  for (int t = 0; t < RT; ++t) fa[t] = *reinterpret_cast<const v4f*>(buf + t * 2048 + a_addr);
  if constexpr (NHWC) {
    fb = *reinterpret_cast<const v4f*>(buf + RT * 2048 + b_addr);
  } else {
    const float* p = reinterpret_cast<const float*>(buf + RT * 2048 + b_addr);
    fb = v4f{p[0], p[64], p[128], p[192]};  // channels k .. k + 3 of this lane's position
  }
}

// One MFMA slot: slot n of a 16-channel chain group is tile n % RT, k-step n / RT (tile-major inside
// a k-step, so consecutive MFMAs never share an accumulator).
template <int RT>
__device__ __forceinline__ void rt_mfma_slot(v4f (&chain)[RT], const v4f (&fa)[RT], const v4f& fb, int n) {
  const int t = n % RT, k = n / RT;
  chain[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t][k], fb[k], k == 0 ? v4f{0.f, 0.f, 0.f, 0.f} : chain[t],
                                                  0, 0, 0);
}
// carry element e of a finished stage: its two chains added in f32, then into f64
template <int RT>
__device__ __forceinline__ void rt_carry(double (&acc)[RT][4], const v4f (&done)[2][RT], int e) {
  acc[e >> 2][e & 3] += (double)(done[0][e >> 2][e & 3] + done[1][e >> 2][e & 3]);
}

template <int RT, int RTMAX, bool NHWC>
__device__ __forceinline__ void rt_block(const RtArgs& a, char* smem, int crop, int t0) {
  constexpr int STAGE = rt_stage_bytes(RT, NHWC);
  constexpr int JOBS = 2 * RT + 8;       // 1 KiB copies per stage: 2 per weight tile, 8 of features
  constexpr int JPW = (JOBS + 3) / 4;    // per wave (upper bound)
  constexpr int R = RT * 16;
  float* Ls = reinterpret_cast<float*>(smem + kRtNbuf * rt_stage_bytes(RTMAX, NHWC));
  float* rowmax = Ls + RTMAX * 16 * kRtLP;
  float* unitmax = rowmax + RTMAX * 16;
  float* bias_s = unitmax + RTMAX * 16;
  int* info_s = reinterpret_cast<int*>(bias_s + RTMAX * 16);
  double* rowsum = reinterpret_cast<double*>(info_s + RTMAX * 16);  // [R][3]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HW = a.H * a.W;
  const int n_stages = a.n_stages;
  const int i16 = lane & 15, g4 = lane >> 4;
  const unsigned lds0 = rt_lds_addr(smem);
  const char* fcrop = reinterpret_cast<const char*>(a.feat) + (size_t)crop * a.C * HW * 4;
  const bool c_tail = (a.C & 31) != 0;
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

  // running (max, sums) of the unit that starts at row tid, across column blocks
  double run_s = 0, run_x = 0, run_y = 0, run_z = 0;
  float run_m = -INFINITY;

  const int n_cb = (HW + 63) >> 6;
  for (int cb = 0; cb < n_cb; ++cb) {
    // ---- this wave's copies: job j = wid + 4 i (0 .. 2RT-1: weight tiles, then 8 feature chunks)
    const char* gptr[JPW];
    unsigned gstride[JPW], voff[JPW], voff_tail[JPW], ldso[JPW];
#pragma unroll
    for (int i = 0; i < JPW; ++i) {
      const int j = min(wid + 4 * i, JOBS - 1);
      if (j < 2 * RT) {
        gptr[i] = a.wt + (size_t)t0 * 2048 + j * 1024;
        gstride[i] = (unsigned)a.n_tiles * 2048u;
        voff[i] = voff_tail[i] = lane * 16;
        ldso[i] = j * 1024;
      } else {
        const int jb = j - 2 * RT;
        gptr[i] = fcrop;
        if constexpr (NHWC) {
          const int pos = jb * 8 + (lane >> 3), slot = (lane & 7) ^ ((pos >> 1) & 7);
          const int P = cb * 64 + pos;
          const unsigned prow = (unsigned)(P < HW ? P : 0) * (unsigned)a.C * 4u;
          gstride[i] = 128;
          voff[i] = prow + slot * 16;
          voff_tail[i] = prow + (c0_last + slot * 4 < a.C ? slot * 16 : 0);
          ldso[i] = RT * 2048 + jb * 1024;
        } else {
          const int ch = jb * 4 + (lane >> 4);
          const int p = cb * 64 + (lane & 15) * 4;
          const unsigned pbytes = (unsigned)(p < HW ? p : 0) * 4u;
          gstride[i] = 32u * (unsigned)HW * 4u;
          voff[i] = (unsigned)ch * (unsigned)HW * 4u + pbytes;
          voff_tail[i] = (unsigned)(c0_last + ch < a.C ? ch : 0) * (unsigned)HW * 4u + pbytes;
          ldso[i] = RT * 2048 + jb * kRtChunkNCHW;
        }
      }
    }
    // (waves whose last job index falls behind the list issue one copy fewer; their vmcnt differs)
    const bool short_wave = wid + 4 * (JPW - 1) >= JOBS;
    auto issue_job = [&](int i, int stage) {
      const unsigned buf = lds0 + (unsigned)(stage & (kRtNbuf - 1)) * STAGE;
      const bool tail = c_tail && stage == n_stages - 1;
      if (i == JPW - 1 && short_wave) return;
      rt_dma16(uniform_ptr(gptr[i]), tail ? voff_tail[i] : voff[i], buf + ldso[i]);
      gptr[i] += gstride[i];
    };
    auto issue = [&](int stage) {
#pragma unroll
      for (int i = 0; i < JPW; ++i) issue_job(i, stage);
    };

    RtRegs<RT> rg;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rg.acc[t][r] = 0.0;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) rg.part[p][q][t] = v4f{0.f, 0.f, 0.f, 0.f};
      rg.ya[t] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    rg.yb = v4f{0.f, 0.f, 0.f, 0.f};

    // prologue: stages 0 .. NBUF-2 in flight
#pragma unroll
    for (int p = 0; p < kRtNbuf - 1; ++p)
      if (p < n_stages) issue(p);

    // Iteration of stage s with parity P = s & 1 (a literal: the partial sets are registers):
    //   wait for this wave's copies of stage s (those of s+1 .. s+NBUF-2 stay in flight), barrier
    //   (all copies of s visible; everyone finished reading the buffer of s-1); read the first
    //   chain's fragments of s; SECOND chain of s-1 from the fragments read before the barrier
    //   (covers the LDS latency), with the copies of stage s+NBUF-1 into the buffer of s-1 and the
    //   second half of stage s-2's carry in the MFMAs' shadow; read the second chain's fragments;
    //   first chain of s with the first half of stage s-1's carry.
    // Every MFMA is followed by at most one carry element (3 VALU) or one copy, and a scheduling
    // fence: a single wave per SIMD issues in order, so work placed between two MFMAs runs in the
    // first one's 32-cycle shadow, while a block of VALU behind a block of MFMAs does not.
#define RT_ITER(S, P)                                                                             \
  {                                                                                               \
    const int s_ = (S);                                                                           \
    const bool more = s_ + kRtNbuf - 1 < n_stages;                                                \
    if (more) {                                                                                   \
      if (short_wave) rt_wait_vmcnt<(kRtNbuf - 2) * (JPW - 1)>();                                 \
      else rt_wait_vmcnt<(kRtNbuf - 2) * JPW>();                                                  \
    } else {                                                                                      \
      rt_wait_vmcnt<0>();                                                                         \
    }                                                                                             \
    __syncthreads();                                                                              \
    const char* buf = smem + (s_ & (kRtNbuf - 1)) * STAGE;                                        \
    v4f xa[RT], xb;                                                                               \
    rt_read_frags<RT, NHWC>(buf, a_off, b_off, xa, xb);                                           \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    _Pragma("unroll") for (int n = 0; n < 4 * RT; ++n) {                                          \
      rt_mfma_slot<RT>(rg.part[(P) ^ 1][1], rg.ya, rg.yb, n);                                     \
      if (n < JPW) {                                                                              \
        if (more) issue_job(n, s_ + kRtNbuf - 1);                                                 \
      } else if ((n - JPW) % 2 == 0 && (n - JPW) / 2 < 2 * RT) {                                  \
        rt_carry<RT>(rg.acc, rg.part[P], 2 * RT + (n - JPW) / 2);                                 \
      }                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    /* (carry elements that found no slot above: JPW + 4RT - 1 > 4RT only for RT = 1) */          \
    _Pragma("unroll") for (int e = (4 * RT - JPW + 1) / 2; e < 2 * RT; ++e)                       \
        rt_carry<RT>(rg.acc, rg.part[P], 2 * RT + e);                                             \
    rt_read_frags<RT, NHWC>(buf, a_off ^ 64, b_q1, rg.ya, rg.yb);                                 \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    _Pragma("unroll") for (int n = 0; n < 4 * RT; ++n) {                                          \
      rt_mfma_slot<RT>(rg.part[P][0], xa, xb, n);                                                 \
      if (n % 2 == 1) rt_carry<RT>(rg.acc, rg.part[(P) ^ 1], n / 2);                              \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
  }
    for (int s = 0; s < n_stages; s += 2) {
      RT_ITER(s, 0)
      if (s + 1 < n_stages) RT_ITER(s + 1, 1)
    }
#undef RT_ITER
    // drain: second chain of the last stage (parity PL) + what is left of the carries
#define RT_DRAIN(PL)                                                                              \
  {                                                                                               \
    _Pragma("unroll") for (int n = 0; n < 4 * RT; ++n) {                                          \
      rt_mfma_slot<RT>(rg.part[PL][1], rg.ya, rg.yb, n);                                          \
      if (n % 2 == 1) rt_carry<RT>(rg.acc, rg.part[(PL) ^ 1], 2 * RT + n / 2);                    \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    _Pragma("unroll") for (int e = 0; e < 4 * RT; ++e) rt_carry<RT>(rg.acc, rg.part[PL], e);      \
  }
    if ((n_stages - 1) & 1) RT_DRAIN(1) else RT_DRAIN(0)
#undef RT_DRAIN

    // ---- logits (+bias) -> LDS.  C/D layout of 16x16x4: col = l & 15, row = 4 (l >> 4) + reg
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = t * 16 + g4 * 4 + r;
        Ls[row * kRtLP + wid * 16 + i16] = (float)(rg.acc[t][r] + (double)bias_s[row]);
      }
    __syncthreads();

    // ---- decode, a 16-lane group per row (RT rounds of 16 rows)
    const int grp = tid >> 4;
    const int pbase = cb * 64 + i16 * 4;  // this lane's 4 positions
    v4f x[RT];
#pragma unroll
    for (int k = 0; k < RT; ++k) {
      const int row = k * 16 + grp;
      x[k] = *reinterpret_cast<const v4f*>(Ls + row * kRtLP + i16 * 4);
      float m = -INFINITY;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (pbase + q < HW) m = fmaxf(m, x[k][q]);
      m = group_max<16>(m);
      if (i16 == 0) rowmax[row] = m;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RT; ++k) {
      const int row = k * 16 + grp;
      const unsigned inf = (unsigned)info_s[row];
      const int kind = inf & 3, d = (inf >> 2) & 0x3fff;
      const int first = kind == 2 ? row - d : row, n = kind == 2 ? a.D : 1;
      float m = -INFINITY;
      for (int kk = i16; kk < n; kk += 16) m = fmaxf(m, rowmax[first + kk]);
      m = group_max<16>(m);
      double s = 0, sx = 0, sy = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = pbase + q;
        if (p < HW && kind != 0) {
          const double e = exp_neg64((double)x[k][q] - (double)m);
          const int h = p / a.W, w = p - h * a.W;
          s += e;
          sx += e * (double)w;
          sy += e * (double)h;
        }
      }
      s = group_sum<16>(s);
      sx = group_sum<16>(sx);
      sy = group_sum<16>(sy);
      if (i16 == 0) {
        rowsum[row * 3 + 0] = s;
        rowsum[row * 3 + 1] = sx;
        rowsum[row * 3 + 2] = sy;
        unitmax[row] = m;
      }
    }
    __syncthreads();
    if (tid < R) {
      const unsigned inf = (unsigned)info_s[tid];
      const int kind = inf & 3, d = (inf >> 2) & 0x3fff, j = (int)(inf >> 16);
      if (kind == 1 || (kind == 2 && d == 0)) {
        const int n = kind == 2 ? a.D : 1;
        double S = 0, SX = 0, SY = 0, SZ = 0;
        for (int k = 0; k < n; ++k) {
          const double s = rowsum[(tid + k) * 3];
          S += s;
          SX += rowsum[(tid + k) * 3 + 1];
          SY += rowsum[(tid + k) * 3 + 2];
          SZ += s * (double)k;
        }
        const float mu = unitmax[tid];
        if (cb == 0) {
          run_m = mu; run_s = S; run_x = SX; run_y = SY; run_z = SZ;
        } else {
          const float mn = fmaxf(run_m, mu);
          const double f1 = exp_neg64((double)run_m - (double)mn), f2 = exp_neg64((double)mu - (double)mn);
          run_s = run_s * f1 + S * f2;
          run_x = run_x * f1 + SX * f2;
          run_y = run_y * f1 + SY * f2;
          run_z = run_z * f1 + SZ * f2;
          run_m = mn;
        }
        if (cb == n_cb - 1) {
          const size_t o = (size_t)crop * a.J + j;
          if (kind == 1) {
            a.c2d[o * 2 + 0] = heatmap_to_px(axis_coord(run_x, run_s, a.W), a.hs);
            a.c2d[o * 2 + 1] = heatmap_to_px(axis_coord(run_y, run_s, a.H), a.hs);
          } else {
            a.c3d[o * 3 + 0] = heatmap_to_mm_xy(axis_coord(run_x, run_s, a.W), a.hs);
            a.c3d[o * 3 + 1] = heatmap_to_mm_xy(axis_coord(run_y, run_s, a.H), a.hs);
            a.c3d[o * 3 + 2] = heatmap_to_mm_z(axis_coord(run_z, run_s, a.D), a.hs);
          }
        }
      }
    }
    // (the next column block's copies only touch the ring, which every wave left before the
    //  barrier behind the logits store; its logits store is many barriers away)
  }
}

template <int RTMAX, bool NHWC>
__global__ __launch_bounds__(256) void head_rt_kernel(RtArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware remap (block id b runs on XCD b % 8): the blocks of a crop share an XCD, so its
  // features come from HBM once and are re-read from that XCD's L2
  const int chunk = 8 * a.n_blocks;
  const int id = blockIdx.x;
  const int crop = (id / chunk) * 8 + (id % 8);
  const int blk = (id % chunk) / 8;
  if (crop >= a.B) return;
  const int t0 = blk * a.rtg;
  const int rt = min(a.rtg, a.n_tiles - t0);
  if (rt == 1) rt_block<1, RTMAX, NHWC>(a, smem, crop, t0);
  if (rt == 2) rt_block<2, RTMAX, NHWC>(a, smem, crop, t0);
  if (rt == 3) rt_block<3, RTMAX, NHWC>(a, smem, crop, t0);
  if constexpr (RTMAX >= 5) {
    if (rt == 4) rt_block<4, RTMAX, NHWC>(a, smem, crop, t0);
    if (rt == 5) rt_block<5, RTMAX, NHWC>(a, smem, crop, t0);
  }
}

template <int RTMAX, bool NHWC>
static int rt_launch_t(const RtArgs& a, hipStream_t stream) {
  constexpr int lds = rt_lds_bytes(RTMAX, NHWC);
  auto kern = head_rt_kernel<RTMAX, NHWC>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  const long long blocks = (long long)((a.B + 7) / 8) * 8 * a.n_blocks;
  if (blocks > 0x7fffffffLL) return MTR_E_SHAPE;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
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

int rt_launch(const float* feat, int layout, const void* section, int B, int C, int H, int W, int J,
              int D, const HeadScale& hs, float* coords2d, float* coords3d_rel, int rtg_hint,
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
  if (g.a == 1 && rtg_hint == 0) {
    // one-tile atoms: 3 tiles per workgroup while that still gives every CU a workgroup; narrower
    // blocks for launches that would otherwise leave CUs idle (few crops, large maps)
    const long long crops8 = (long long)((B + 7) / 8) * 8;
    rtg_hint = 3;
    while (rtg_hint > 1 && crops8 * ((g.n_tiles + rtg_hint - 1) / rtg_hint) < 256) --rtg_hint;
  }
  a.rtg = rt_block_tiles(g, rtg_hint);
  a.n_blocks = (g.n_tiles + a.rtg - 1) / a.rtg;
  a.hs = hs;
  a.c2d = coords2d;
  a.c3d = coords3d_rel;
  const bool nhwc = layout == MTR_NHWC;
  if (a.rtg <= 3)
    return nhwc ? rt_launch_t<3, true>(a, stream) : rt_launch_t<3, false>(a, stream);
  return nhwc ? rt_launch_t<5, true>(a, stream) : rt_launch_t<5, false>(a, stream);
}

}  // namespace mtr
