// K1 + K2-K4 fused, 16-bit features, EIGHT waves in two alternating halves (round 6): the joint-group head
// kernel for the wide tiles (configs[4]: J = 122 on 12x12 maps -- 18 joint groups x 5 column tiles per crop).
//
// Replaces MetrabsHeads.forward (metrabs_pytorch/models/metrabs.py:75-85) like the kernels of head_fused.hip,
// whose LDS tile formats, packed weights, MFMA order per accumulator and decode epilogue it shares -- the
// results are bit-identical (tests/test_gpu_head.py compares every dispatch choice with torch.equal).
//
// Why another variant (profiles/r05q_head16_cycle_trace.jsonl).  In head_fused16dma_kernel a
// wave's stage is a serial chain -- wait for its copies (320 cycles), barrier (160), issue the next stage's nine
// 1 KiB copy pieces (690: the wave issues nothing else meanwhile), fragment reads + 24 MFMAs (1,350, of which 768
// are matrix pipe) -- and only a SECOND workgroup on the CU overlaps any of it: the matrix pipe is 41 % busy at
// 256 crops, and at 32 crops (one workgroup per CU) a stage of 2,700 cycles holds 768 of MFMA.  A sixth of those
// MFMAs multiply the padding tile of the odd column-tile count, and a workgroup of two joint groups pulls 36 KiB
// through the CU's vector-memory path per 80 useful MFMAs.  Here
//   * a workgroup is FOUR joint groups (256 rows) x all column tiles of one crop: 32 + 20 KiB per 160 MFMAs
//     (-28 % bytes per MFMA), no padding-tile MFMAs: wave w owns the 32-row block w against every column tile
//     (CT accumulator tiles, one weight fragment read per CT MFMAs);
//   * the eight waves are two halves, X = waves 0-3 (groups 0, 1) and Y = waves 4-7 (groups 2, 3) -- one wave of
//     each half per SIMD.  Every 64-channel stage has two phases separated by barriers: in phase 0 X runs its
//     4 CT MFMAs of the stage while Y issues copy pieces of the NEXT stage (all feature pieces + X's weight
//     rows); in phase 1 they swap (X issues Y's weight rows).  On every SIMD one wave owns the matrix pipe while
//     its partner sits in the vector-memory issue path -- the two long serial parts of the old stage now run
//     side by side inside ONE workgroup, whatever the launch size;
//   * a wave waits for its own copies (s_waitcnt vmcnt(0)) at the END of its compute phase, i.e. every copy has at
//     least a whole phase to land, and the barrier behind it publishes them to the other half.
// LDS: two stage buffers of 32 KiB weights + 4 CT KiB features (106,496 B at CT = 5): one workgroup per CU.
#include "common.h"
#include "head16.h"

namespace mtr {

constexpr int kPpGroups = kPpGroupsPerWorkgroup;   // joint groups per workgroup (4)
constexpr int kPpRows = kPpGroups * kRows;         // 256 tile rows
constexpr int kPpAStage = kPpRows * 128;           // bytes of a weight stage in LDS

#ifndef MTR_PP_TRACE
#define MTR_PP_TRACE 0    // developer builds (tools/experiments/head16_pp_trace.py): cycle stamps (s_memtime) of the phases of
                          // every stage, summed per wave; the workgroups whose blockIdx % 293 == 0 write theirs BEHIND the
                          // launch's coords2d (the probe passes a longer buffer): 128 floats per workgroup, 16 per wave
#endif
#ifndef MTR_PP_SAMECROP
#define MTR_PP_SAMECROP 0
#endif
#ifndef MTR_PP_ABLATE
#define MTR_PP_ABLATE 0   // developer-only timing ablations (tools/experiments/head16_pp_probe.py): 1 = no decode,
                          // 2 = no logits store + no decode, 4 = no MFMA, 8 = no copies in the K loop, 16 = no fragment reads
#endif

template <typename FeatT, int CT, bool NHWC>
__global__ __launch_bounds__(512, 2) void head_fused16pp_kernel(
    const FeatT* __restrict__ feat, const float* __restrict__ packed, int B, int C, int H, int W, int J, int D,
    HeadGeom g, HeadScale hs, float* __restrict__ coords2d, float* __restrict__ coords3d_rel) {
  constexpr int HWP = hw_pad32<CT>();
  constexpr int B_STAGE = CT * 32 * 128;          // bytes
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* As = reinterpret_cast<char*>(smem);       // [2][256][128 B]
  char* Bs = As + 2 * kPpAStage;                  // [2][CT*32][128 B]   (NCHW: [64 ch][HW * 2 B] inside it)
  float* Ls = smem;                               // epilogue alias: [128][HWP], two groups at a time

  const int HW = H * W;
  const int wg_per_crop = (g.n_groups + kPpGroups - 1) / kPpGroups;
  const int chunk = 8 * wg_per_crop;
  const int id = blockIdx.x;
  const int crop = (id / chunk) * 8 + (id % 8);   // (the workgroups of a crop on one XCD, as head_fused.hip)
  const int grp0 = ((id % chunk) / 8) * kPpGroups;
  if (crop >= B) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0 .. 7 = this wave's 32-row block
  const bool is_y = wid >= 4;                                 // (wave-uniform)
  const int wi = wid & 3;
  const bool y_active = grp0 + 2 < g.n_groups;                // the workgroup's second pair of groups exists
  const int n_st = C / kKH;
  const float* bias = packed;
  const FeatT* w16 = reinterpret_cast<const FeatT*>(packed + (size_t)g.n_groups * kRows);
#if MTR_PP_SAMECROP   // (developer timing probe: every workgroup reads crop 0's features -- all L2 hits; the poses are garbage)
  const FeatT* fcrop = feat;
#else
  const FeatT* fcrop = feat + (size_t)crop * C * HW;
#endif
  const int fi = lane & 31, fg = lane >> 5;

  // NHWC rows of positions >= HW stay zero (masked copy lanes)
  for (int v = tid; v < 2 * B_STAGE / 16; v += 512)
    reinterpret_cast<v4u*>(Bs)[v] = v4u{0u, 0u, 0u, 0u};

  // ---- copy pieces (1 KiB = 8 tile rows per wave-wide global_load_lds_dwordx4; lane L -> LDS base + 16 L).
  // Weight piece p covers tile rows 8 p .. 8 p + 7 (p < 16: X's rows, else Y's); lane (lr, ls) fetches channel slot
  // ls ^ swz(row) of row 8 p + lr (the XOR swizzle on the source side, head_fused16dma_kernel).  A wave's
  // pieces: Y wave wi issues feature pieces 4 i + wi (i < CT) and X's weight pieces 4 i + wi (i < 4); X wave wi
  // issues Y's weight pieces 16 + 4 i + wi (i < 4).
  const int lr = lane >> 3, ls = lane & 7;
  const FeatT* a_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = ((is_y ? 0 : 16) + i * 4 + wi) * 8 + lr;          // tile row 0 .. 255
    const int grp = min(grp0 + (row >> 6), g.n_groups - 1);
    a_src[i] = w16 + (size_t)grp * n_st * (kRows * kKH) + (row & 63) * kKH + ((ls ^ swz(row)) << 3);
  }
  const FeatT* b_src[CT];
  bool b_on[CT];
  const int n_chunks = HW >> 3;
  const size_t b_stage_elems = NHWC ? (size_t)kKH : (size_t)kKH * HW;
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    const int piece = i * 4 + wi;
    if constexpr (NHWC) {
      const int pos = piece * 8 + lr;
      b_on[i] = pos < HW;
      b_src[i] = fcrop + (size_t)(b_on[i] ? pos : 0) * C + ((ls ^ swz(pos)) << 3);
    } else {  // the tile keeps the memory layout [channel][position]; chunk rotation: head_fused16dma_kernel
      const int cid = piece * 64 + lane;
      b_on[i] = cid < kKH * n_chunks;
      const int k = b_on[i] ? cid / n_chunks : 0, jl = b_on[i] ? cid - k * n_chunks : 0;
      const int rot = nchw_chunk_rot(k, n_chunks);
      const int j = jl >= rot ? jl - rot : jl - rot + n_chunks;
      b_src[i] = fcrop + (size_t)k * HW + j * 8;
    }
  }
  const unsigned As_a = lds_byte_addr(As), Bs_a = lds_byte_addr(Bs);
  auto issue = [&](int stage, int buf) {   // this wave's share of a stage
    if (is_y) {
#pragma unroll
      for (int i = 0; i < CT; ++i)
        if (b_on[i])
          dma16_to_lds_asm(b_src[i] + (size_t)stage * b_stage_elems, Bs_a + buf * B_STAGE + (i * 4 + wi) * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        dma16_to_lds_asm(a_src[i] + (size_t)stage * (kRows * kKH), As_a + buf * kPpAStage + (i * 4 + wi) * 1024);
    } else if (y_active) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        dma16_to_lds_asm(a_src[i] + (size_t)stage * (kRows * kKH), As_a + buf * kPpAStage + (16 + i * 4 + wi) * 1024);
    }
  };

  // ---- fragment addresses: lane (fi, fg) of step u reads slot 2 u + fg of tile row 32 wid + fi (weights) and of
  // position 32 t + fi (features; NCHW: the transposing read of head_fused16dma_kernel)
  int a_off;
  {
    const int row = wid * 32 + fi;
    a_off = row * 128 + ((fg ^ swz(row)) << 4);
  }
  int b_off[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    if constexpr (NHWC) {
      const int pos = t * 32 + fi;
      b_off[t] = pos * 128 + ((fg ^ swz(pos)) << 4);
    } else {
      const int G = lane >> 4, r = lane & 15, q = r & 3, ci = r >> 2;
      const int P = t * 32 + 16 * (G & 1) + 4 * q;
      const int Pc = P < HW ? P : 0;
      int jl = (Pc >> 3) + nchw_chunk_rot(ci, n_chunks);
      jl = jl >= n_chunks ? jl - n_chunks : jl;
      b_off[t] = (8 * fg + ci) * (HW * 2) + jl * 16 + (Pc & 7) * 2;
    }
  }
  const int tr_pitch4 = 4 * HW * 2;

  f32x16 acc[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) acc[t] = f32x16{0};

  // this wave's 4 CT MFMAs of a stage.  The fragments of step u + 1 are REQUESTED before the MFMAs of step u are
  // issued (two register sets; __builtin_amdgcn_sched_barrier keeps the compiler from sinking the reads back down):
  // left to itself the compiler issued every ds_read one or two MFMAs ahead of its use and drained lgkmcnt at every
  // step -- 20 MFMAs took ~1,650 cycles of a phase instead of 640 (the first build of this kernel, 200 vs 168 us at
  // 256 crops; profiles/r06b_head16_pp.jsonl).  Same MFMAs in the same order per accumulator.
  auto read_frags = [&](const char* Ab, const char* Bb, int u, v4u& af, v4u (&bf)[CT]) {
    af = (MTR_PP_ABLATE & 16) ? v4u{(unsigned)a_off, 1u, 2u, (unsigned)u}
                              : *reinterpret_cast<const v4u*>(Ab + (a_off ^ (u << 5)));
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      if (MTR_PP_ABLATE & 16) {
        bf[t] = v4u{(unsigned)b_off[t], 1u, 2u, (unsigned)u};
      } else if constexpr (NHWC) {
        bf[t] = *reinterpret_cast<const v4u*>(Bb + (b_off[t] ^ (u << 5)));
      } else {
        const char* p = Bb + b_off[t] + u * (4 * tr_pitch4);
        bf[t] = lds_read_tr16_pair(p, p + tr_pitch4);
      }
    }
  };
  auto compute = [&](int st) {
    const char* Ab = As + (st & 1) * kPpAStage;
    const char* Bb = Bs + (st & 1) * B_STAGE;
    v4u af[2], bf[2][CT];
    read_frags(Ab, Bb, 0, af[0], bf[0]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u + 1 < 4) read_frags(Ab, Bb, u + 1, af[(u + 1) & 1], bf[(u + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        if (MTR_PP_ABLATE & 4) acc[t][0] += __builtin_bit_cast(float, af[u & 1][0] ^ bf[u & 1][t][0]);
        else acc[t] = Mfma16<FeatT>::run(af[u & 1], bf[u & 1][t], acc[t]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

#if MTR_PP_TRACE
  unsigned long long tr_t = __builtin_amdgcn_s_memtime(), tr_start = tr_t;
  unsigned tr_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MTR_PP_STAMP(K)                                        \
  {                                                            \
    const unsigned long long now = __builtin_amdgcn_s_memtime(); \
    tr_sum[K] += (unsigned)(now - tr_t);                       \
    tr_t = now;                                                \
  }
#else
#define MTR_PP_STAMP(K)
#endif
  __syncthreads();  // zero fill done
  issue(0, 0);      // (X's rows and the features by the Y waves, Y's rows by the X waves: the steady-state shares)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MTR_PP_STAMP(6)   // prologue: zero fill, addresses, stage 0
  for (int st = 0; st < n_st; ++st) {
    const bool more = st + 1 < n_st && !(MTR_PP_ABLATE & 8);
    // ---- phase 0: X multiplies stage st; Y issues the features and X's rows of stage st + 1 into the other
    // buffer (last read in phase 1 of stage st - 1 by Y, phase 0 by X)
    if (!is_y) {
      compute(st);
      MTR_PP_STAMP(0)   // X: reads + MFMAs issued;  Y: copies issued
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // Y's rows of stage st (issued in phase 1 of st - 1)
    } else if (more) {
      issue(st + 1, (st + 1) & 1);
      MTR_PP_STAMP(0)
    }
    MTR_PP_STAMP(1)     // X: its wait for copies
    __syncthreads();
    MTR_PP_STAMP(2)     // barrier
    // ---- phase 1: Y multiplies stage st; X issues Y's rows of stage st + 1
    if (is_y) {
      if (y_active) compute(st);
      MTR_PP_STAMP(3)   // Y: reads + MFMAs issued;  X: copies issued
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // features + X's rows of stage st + 1
    } else if (more) {
      issue(st + 1, (st + 1) & 1);
      MTR_PP_STAMP(3)
    }
    MTR_PP_STAMP(4)     // Y: its wait for copies
    __syncthreads();
    MTR_PP_STAMP(5)     // barrier
  }

  // ---- epilogue: two joint groups at a time through LDS [128][HWP] (written by the half that owns them), decoded by
  // four waves each -- the shipped kernels' decode on the same logits, joint for joint
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h == 1) __syncthreads();   // the first pair's logits are no longer read
    if (grp0 + 2 * h >= g.n_groups) break;   // (uniform over the workgroup)
    if ((wid >> 2) == h && !(MTR_PP_ABLATE & 2)) {
      const int grp = min(grp0 + (wid >> 1), g.n_groups - 1);
      const float* bgrp = bias + (size_t)grp * kRows + (wid & 1) * 32;
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 8 * (r >> 2) + 4 * fg + (r & 3);          // inside this wave's 32-row block
          Ls[(wi * 32 + row) * HWP + t * 32 + fi] = acc[t][r] + bgrp[row];
        }
    }
    __syncthreads();
    if (MTR_PP_ABLATE & 3) {  // no decode: one store per workgroup keeps the GEMM alive
      if (tid == 0) coords2d[(size_t)crop * J * 2 + (grp0 + 2 * h)] = acc[0][0] + Ls[0];
      continue;
    }
    const int grp = grp0 + 2 * h + (wid >> 2);   // waves 0-3: the pair's first group, waves 4-7: its second
    if (grp < g.n_groups)
      decode_group_from_lds<false, (CT > 2 ? 4 : 2)>(Ls + (size_t)(wid >> 2) * kRows * HWP, HWP, grp, g, crop, J, D, H, W,
                                                     hs, coords2d, coords3d_rel, wi, lane);
  }
#if MTR_PP_TRACE
  MTR_PP_STAMP(7)   // epilogue
  if (id % 293 == 0 && lane == 0) {
    float* o = coords2d + (size_t)B * J * 2 + (size_t)(id / 293) * 128 + wid * 16;   // (BEHIND the launch's own outputs)
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (float)tr_sum[k];
    o[8] = (float)(unsigned)(tr_t - tr_start);
    o[9] = (float)id;
    o[10] = (float)n_st;
  }
#endif
}

template <int CT>
constexpr size_t head16_pp_lds_bytes() {
  constexpr size_t stage = 2 * ((size_t)kPpAStage + (size_t)CT * 32 * 128);
  constexpr size_t logits = (size_t)2 * kRows * hw_pad32<CT>() * sizeof(float);
  return stage > logits ? stage : logits;
}

template <typename FeatT, int CT, bool NHWC>
static int launch_pp(const void* feat, const float* packed, int B, int C, int H, int W, int J, int D, const HeadGeom& g,
                     const HeadScale& hs, float* c2d, float* c3d, hipStream_t stream) {
  constexpr size_t lds = head16_pp_lds_bytes<CT>();
  const int chunk = 8 * ((g.n_groups + kPpGroups - 1) / kPpGroups);
  const long long blocks = (long long)((B + 7) / 8) * chunk;
  if (blocks > 0x7fffffffLL) return MTR_E_SHAPE;
  auto kern = head_fused16pp_kernel<FeatT, CT, NHWC>;
  const int rc = allow_dynamic_lds((const void*)kern, lds);
  if (rc != MTR_OK) return rc;
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, stream, (const FeatT*)feat, packed, B, C, H, W, J, D,
                     g, hs, c2d, c3d);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

template <typename FeatT, bool NHWC>
static int pp_by_tiles(int ct, const void* feat, const float* packed, int B, int C, int H, int W, int J, int D,
                       const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d, hipStream_t stream) {
  switch (ct) {
    case 3: return launch_pp<FeatT, 3, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    case 4: return launch_pp<FeatT, 4, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    case 5: return launch_pp<FeatT, 5, NHWC>(feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    default: return MTR_E_SHAPE;
  }
}

bool head16_pp_supported(int C, int H, int W, int layout) {
  const int hw = H * W, ct = (hw + 31) / 32;
  return ct >= 3 && ct <= 5 && C % kKH == 0 && (layout == MTR_NHWC || (hw % 8 == 0 && hw >= 64));
}

int head16_pp_launch(int feat_dtype, int layout, const void* feat, const float* packed, int B, int C, int H, int W,
                     int J, int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d, hipStream_t stream) {
  if (!head16_pp_supported(C, H, W, layout)) return MTR_E_SHAPE;
  const int ct = (H * W + 31) / 32;
  if (feat_dtype == MTR_F16) {
    if (layout == MTR_NHWC) return pp_by_tiles<__half, true>(ct, feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    return pp_by_tiles<__half, false>(ct, feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  }
  if (layout == MTR_NHWC)
    return pp_by_tiles<__hip_bfloat16, true>(ct, feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  return pp_by_tiles<__hip_bfloat16, false>(ct, feat, packed, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
}

}  // namespace mtr
