// K1 + K2-K4 fused, 16-bit features, WEIGHTS IN REGISTERS (round 5): the joint-group head kernel for the
// wide tiles (configs[4]: J = 122 on 12x12 maps -- 18 joint groups x 5 column tiles per crop).
//
// Replaces MetrabsHeads.forward (metrabs_pytorch/models/metrabs.py:75-85) like the kernels of
// head_fused.hip, whose stages, MFMA order per accumulator and decode epilogue it shares -- the results
// are bit-identical (tests/test_gpu_head.py compares every dispatch choice with torch.equal).
//
// Why another variant.  head_fused16dma_kernel stages BOTH operands through LDS: per 64-channel stage a
// workgroup of two joint groups pulls 16 KiB of weights + 20 KiB of features through the CU's 64 B/clk vector
// memory path for 96 MFMAs (a sixth of them on the padding tile of the odd column-tile count): the copy pieces'
// issue slots are as long as the wave's MFMAs, and round 4 measured every unit a third busy.  Here
//   * a WAVE owns a whole joint group (64 rows = two 32-row MFMA blocks) against ALL column tiles of the
//     crop: 10 accumulator tiles, every feature fragment read from LDS feeds two MFMAs and no MFMA
//     multiplies a padding tile;
//   * its weight fragments are what one lane group needs and nobody else: every lane loads its own 16 bytes
//     per 16-channel step straight into registers from the FRAGMENT-MAJOR section of the packed blob
//     ([group][stage][row block][step][lane][8 channels]: a wave-wide load is 1 KiB contiguous = 8 cache lines;
//     from the row-major tiles the same load touched 32 lines and the variant was 12 % SLOWER than the shipped
//     kernel, profiles/r05d_*), re-loaded for the next stage right behind the step that consumed them -- no
//     LDS copy, no LDS read, no copy pieces;
//   * only the features go through LDS (NCHW needs the transposing read; all waves share them): 20 KiB per
//     stage and workgroup whatever the number of groups, two buffers;
//   * GPW = 2 ... 4 groups (waves) per workgroup: at 4 the workgroup pulls 32 + 20 KiB per 160 MFMAs.
#include "common.h"
#include "head16.h"

namespace mtr {

// 16 bytes per lane from global memory into registers, issued from inline asm (asynchronous: the caller waits
// with s_waitcnt vmcnt and ties the register to that wait).  OFF: immediate byte offset, 0 .. 4095.
template <int OFF>
__device__ __forceinline__ v4u gload16_asm(const void* p) {
  v4u r;
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(r) : "v"(p), "n"(OFF));
  return r;
}

template <typename FeatT, int CT, bool NHWC, int GPW>
__global__ __launch_bounds__(64 * GPW, 2) void head_fused16areg_kernel(
    const FeatT* __restrict__ feat, const float* __restrict__ bias, const FeatT* __restrict__ wfrag, int B, int C,
    int H, int W, int J, int D, HeadGeom g, HeadScale hs, float* __restrict__ coords2d,
    float* __restrict__ coords3d_rel) {
  constexpr int NW = GPW;                         // waves: one per joint group
  constexpr int NT = 64 * NW;
  constexpr int HWP = hw_pad32<CT>();
  constexpr int B_STAGE = CT * 32 * 128;          // bytes
  constexpr int NPIECE = 4 * CT;                  // 1 KiB copy pieces per stage
  constexpr int PPW = (NPIECE + NW - 1) / NW;     // ... per wave (the last ones of some waves: none)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* Bs = reinterpret_cast<char*>(smem);       // [2][CT*32][128 B]
  float* Ls = smem;                               // epilogue alias: [64][HWP], one group at a time

  const int HW = H * W;
  const int wg_per_crop = (g.n_groups + GPW - 1) / GPW;
  const int chunk = 8 * wg_per_crop;
  const int id = blockIdx.x;
  const int crop = (id / chunk) * 8 + (id % 8);   // (the joint groups of a crop on one XCD, as head_fused.hip)
  const int grp0 = ((id % chunk) / 8) * GPW;
  if (crop >= B) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_st = C / kKH;
  const FeatT* fcrop = feat + (size_t)crop * C * HW;
  const int fi = lane & 31, fg = lane >> 5;

  // NHWC rows of positions >= HW stay zero (masked copy lanes); NCHW needs no fill (the padding columns read
  // valid data of the crop), but the fill is 2 x 20 KiB once per workgroup
  for (int v = tid; v < 2 * B_STAGE / 16; v += NT)
    reinterpret_cast<v4u*>(Bs)[v] = v4u{0u, 0u, 0u, 0u};

  // ---- features: the copy pieces of head_fused16dma_kernel, dealt to NW waves (piece p = i * NW + wave)
  const int lr = lane >> 3, ls = lane & 7;
  const FeatT* b_src[PPW];
  bool b_on[PPW];
  const int n_chunks = HW >> 3;
  const size_t b_stage_elems = NHWC ? (size_t)kKH : (size_t)kKH * HW;
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int piece = i * NW + wid;
    if constexpr (NHWC) {
      const int pos = piece * 8 + lr;
      b_on[i] = piece < NPIECE && pos < HW;
      b_src[i] = fcrop + (size_t)(b_on[i] ? pos : 0) * C + ((ls ^ swz(pos)) << 3);
    } else {
      const int cid = piece * 64 + lane;  // linear 16-byte chunk of the stage in LDS
      b_on[i] = piece < NPIECE && cid < kKH * n_chunks;
      const int k = b_on[i] ? cid / n_chunks : 0, jl = b_on[i] ? cid - k * n_chunks : 0;
      const int rot = nchw_chunk_rot(k, n_chunks);
      const int j = jl >= rot ? jl - rot : jl - rot + n_chunks;  // source chunk of LDS chunk jl
      b_src[i] = fcrop + (size_t)k * HW + j * 8;
    }
  }
  int b_off[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    if constexpr (NHWC) {
      const int pos = t * 32 + fi;
      b_off[t] = pos * 128 + ((fg ^ swz(pos)) << 4);
    } else {  // ds_read_b64_tr_b16 addressing: see head_fused16dma_kernel
      const int G = lane >> 4, r = lane & 15, q = r & 3, ci = r >> 2;
      const int P = t * 32 + 16 * (G & 1) + 4 * q;
      const int Pc = P < HW ? P : 0;
      int jl = (Pc >> 3) + nchw_chunk_rot(ci, n_chunks);
      jl = jl >= n_chunks ? jl - n_chunks : jl;
      b_off[t] = (8 * fg + ci) * (HW * 2) + jl * 16 + (Pc & 7) * 2;
    }
  }
  const int tr_pitch4 = 4 * HW * 2;

  f32x16 acc[2][CT];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[k][t] = f32x16{0};

  const unsigned Bs_a = lds_byte_addr(Bs);
  auto issue = [&](int stage, int buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      if (b_on[i])
        dma16_to_lds_asm(b_src[i] + (size_t)stage * b_stage_elems, Bs_a + buf * B_STAGE + (i * NW + wid) * 1024);
  };
  __syncthreads();  // zero fill done
  issue(0, 0);
  // ---- weights: lane L of (row block k, step u) of a stage holds the 16 bytes at [k][u][L] of the stage's
  // 8 KiB fragment-major block (a workgroup behind the last group of the crop computes that group again and
  // stores nothing).  The loads are inline asm: the compiler sinks plain loads to the top of the NEXT iteration
  // (seen in the ISA: all eight right in front of the stage barrier, their latency exposed every stage) and
  // counts only its own loads when it waits.  Issued here, waited for by hand: step u's two fragments of stage
  // st + 1 go out right behind the MFMAs of step u of stage st, into the registers those just read.  Queue
  // (oldest first) at the top of a stage: [copies of this stage][A0 A1 A2 A3 of this stage, two loads each] ->
  // vmcnt(2) leaves only A3 in flight; in front of step 3: [A3][copies of the next stage][A0' A1' A2'] ->
  // vmcnt(6).  The "+v" operands tie the registers to the wait.
  const int my_grp = min(grp0 + wid, g.n_groups - 1);
  const char* ap0 = reinterpret_cast<const char*>(wfrag + (size_t)my_grp * n_st * (kRows * kKH)) + lane * 16;
  const char* ap1 = ap0 + 4096;   // row block 1
  v4u a[2][4];
#define MTR_AREG_LOAD(U)                  \
  a[0][U] = gload16_asm<1024 * (U)>(ap0); \
  a[1][U] = gload16_asm<1024 * (U)>(ap1);
  MTR_AREG_LOAD(0) MTR_AREG_LOAD(1) MTR_AREG_LOAD(2) MTR_AREG_LOAD(3)
  for (int st = 0; st < n_st; ++st) {
    asm volatile("s_waitcnt vmcnt(2)"
                 : "+v"(a[0][0]), "+v"(a[1][0]), "+v"(a[0][1]), "+v"(a[1][1]), "+v"(a[0][2]), "+v"(a[1][2])
                 :
                 : "memory");  // this wave's copies of stage st have landed, and its weights of steps 0 .. 2
    __syncthreads();  // ... everyone's copies; and every wave has consumed the fragments of stage st - 1
    const int cur = st & 1;
    if (st + 1 < n_st) {  // (wave-uniform)
      issue(st + 1, cur ^ 1);
      ap0 += kRows * kKH * 2;
      ap1 += kRows * kKH * 2;
    }
    const char* Bb = Bs + cur * B_STAGE;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u == 3) asm volatile("s_waitcnt vmcnt(6)" : "+v"(a[0][3]), "+v"(a[1][3]));
      v4u bf[CT];
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        if constexpr (NHWC) {
          bf[t] = *reinterpret_cast<const v4u*>(Bb + (b_off[t] ^ (u << 5)));
        } else {
          const char* p = Bb + b_off[t] + u * (4 * tr_pitch4);
          bf[t] = lds_read_tr16_pair(p, p + tr_pitch4);
        }
      }
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k) acc[k][t] = Mfma16<FeatT>::run(a[k][u], bf[t], acc[k][t]);
      // this step's weights of the NEXT stage, into the registers just consumed: a whole stage in flight.
      // Behind the last stage the same loads are a repeat of that stage -- the wait counts stay the same in
      // every iteration (a branch around a wait makes the compiler copy the tied registers in front of it:
      // seen in the ISA) -- and the wait behind the loop keeps the registers theirs until they have landed.
      if (u == 0) { MTR_AREG_LOAD(0) }
      if (u == 1) { MTR_AREG_LOAD(1) }
      if (u == 2) { MTR_AREG_LOAD(2) }
      if (u == 3) { MTR_AREG_LOAD(3) }
    }
  }
#undef MTR_AREG_LOAD
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(a[0][0]), "+v"(a[1][0]), "+v"(a[0][1]), "+v"(a[1][1]), "+v"(a[0][2]), "+v"(a[1][2]),
                 "+v"(a[0][3]), "+v"(a[1][3]));

  // ---- epilogue: one joint group at a time through LDS [64][HWP] (written by its wave), decoded by all waves
#pragma unroll
  for (int q = 0; q < GPW; ++q) {
    __syncthreads();
    if (grp0 + q >= g.n_groups) break;
    if (wid == q) {
      const float* bgrp = bias + (size_t)(grp0 + q) * kRows;
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = k * 32 + 8 * (r >> 2) + 4 * fg + (r & 3);
            Ls[row * HWP + t * 32 + fi] = acc[k][t][r] + bgrp[row];
          }
    }
    __syncthreads();
    decode_group_from_lds<false, (CT > 2 ? 4 : 2), NW, false>(Ls, HWP, grp0 + q, g, crop, J, D, H, W, hs, coords2d,
                                                        coords3d_rel, wid, lane);
  }
}

template <int CT>
constexpr size_t head16_areg_lds_bytes() {
  constexpr size_t stage = 2 * (size_t)CT * 32 * 128;
  constexpr size_t logits = (size_t)kRows * hw_pad32<CT>() * sizeof(float);
  return stage > logits ? stage : logits;
}

template <typename FeatT, int CT, bool NHWC, int GPW>
static int launch_areg(const void* feat, const float* bias, const void* wfrag, int B, int C, int H, int W, int J,
                       int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d, hipStream_t stream) {
  constexpr size_t lds = head16_areg_lds_bytes<CT>();
  const int chunk = 8 * ((g.n_groups + GPW - 1) / GPW);
  const long long blocks = (long long)((B + 7) / 8) * chunk;
  if (blocks > 0x7fffffffLL) return MTR_E_SHAPE;
  auto kern = head_fused16areg_kernel<FeatT, CT, NHWC, GPW>;
  if (lds > 64 * 1024) {
    const int rc = allow_dynamic_lds((const void*)kern, lds);
    if (rc != MTR_OK) return rc;
  }
  MTR_CLEAR_STALE();
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * GPW), lds, stream, (const FeatT*)feat, bias,
                     (const FeatT*)wfrag, B, C, H, W, J, D, g, hs, c2d, c3d);
  MTR_CHECK_LAUNCH();
  return MTR_OK;
}

template <typename FeatT, int CT, bool NHWC>
static int areg_by_groups(int gpw, const void* feat, const float* bias, const void* wfrag, int B, int C, int H, int W,
                          int J, int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d,
                          hipStream_t stream) {
  switch (gpw) {
    case 2: return launch_areg<FeatT, CT, NHWC, 2>(feat, bias, wfrag, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    case 3: return launch_areg<FeatT, CT, NHWC, 3>(feat, bias, wfrag, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    case 4: return launch_areg<FeatT, CT, NHWC, 4>(feat, bias, wfrag, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    default: return MTR_E_PARAM;
  }
}

template <typename FeatT, bool NHWC>
static int areg_by_tiles(int ct, int gpw, const void* feat, const float* bias, const void* wfrag, int B, int C, int H,
                         int W, int J, int D, const HeadGeom& g, const HeadScale& hs, float* c2d, float* c3d,
                         hipStream_t stream) {
  switch (ct) {
    case 3: return areg_by_groups<FeatT, 3, NHWC>(gpw, feat, bias, wfrag, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    case 4: return areg_by_groups<FeatT, 4, NHWC>(gpw, feat, bias, wfrag, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    case 5: return areg_by_groups<FeatT, 5, NHWC>(gpw, feat, bias, wfrag, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    default: return MTR_E_SHAPE;
  }
}

bool head16_areg_supported(int C, int H, int W, int layout) {
  const int hw = H * W, ct = (hw + 31) / 32;
  return ct >= 3 && ct <= 5 && C % kKH == 0 && (layout == MTR_NHWC || (hw % 8 == 0 && hw >= 64));
}

int head16_areg_launch(int feat_dtype, int layout, int gpw, const void* feat, const float* bias, const void* wfrag,
                       int B, int C, int H, int W, int J, int D, const HeadGeom& g, const HeadScale& hs, float* c2d,
                       float* c3d, hipStream_t stream) {
  if (!head16_areg_supported(C, H, W, layout)) return MTR_E_SHAPE;
  const int ct = (H * W + 31) / 32;
  if (feat_dtype == MTR_F16) {
    if (layout == MTR_NHWC)
      return areg_by_tiles<__half, true>(ct, gpw, feat, bias, wfrag, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
    return areg_by_tiles<__half, false>(ct, gpw, feat, bias, wfrag, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  }
  if (layout == MTR_NHWC)
    return areg_by_tiles<__hip_bfloat16, true>(ct, gpw, feat, bias, wfrag, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
  return areg_by_tiles<__hip_bfloat16, false>(ct, gpw, feat, bias, wfrag, B, C, H, W, J, D, g, hs, c2d, c3d, stream);
}

}  // namespace mtr
