// Implicit-GEMM Conv2D/Dense forward, weight-gradient and input-gradient kernels for
// gfx950 (MI355X, CDNA4), fp32 in / fp32 accumulate on the matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, bitwise an fmaf chain, 157 TFLOP/s dense peak).
//
// All three kernels share one LDS tile core: As[kk][i] and Bs[kk][j] (kk = reduction
// index, 32 per step), each wave owning 32x32 output tiles fed by ds_read_b32 that are
// bank-conflict free (consecutive lanes -> consecutive i / j).  What differs is how the
// operand tiles are gathered from HBM:
//   fwd   : A = im2col(X)[m,k] (NHWC gather, uint8->f32 fused, minibatch row gather
//           fused), B = W[k,n]                      -> Y[m,n] = act(. + bias)
//   wgrad : A = im2col(X)^T[k,m], B = dY[m,n]       -> dW[k,n] (+ db), split over m
//   dgrad : A = dY gathered per stride-parity class, B = W^T -> dX * act'(X)
// Fetches are 16-byte (4-byte for uint8) vectors along the contiguous NHWC channel
// axis; operands whose contiguous axis is the reduction axis are transposed on the way
// into LDS (row stride == 1 mod 32 words, conflict-free ds_write_b32).
//
// Replaces the TensorFlow Conv2D/Dense forward+backward ops the reference runs inside
// sess.run (xt/model/ppo/ppo.py:129, xt/model/impala/impala_cnn_opt.py:255).
#include <stdlib.h>
#include "xt_common.h"
#include "xt_igemm.h"
#include "xt_heads_dev.h"
#include "xt_direct_dev.h"
#include "xt_conv1_dev.h"

namespace xt {

__device__ __forceinline__ float4 sel4(bool ok, float4 v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// Raw 4-element input group as it comes from HBM: one dword (4 x uint8) or four floats.  The uint8->f32
// transform and the zero-select for padding/tails are applied when the group is written to LDS, NOT when it
// is loaded, so that the global loads of step s+2 stay in flight behind the MFMAs of step s.
template <bool U8> struct Raw4;
template <> struct Raw4<true> { uint32_t v; };
template <> struct Raw4<false> { float4 v; };

template <bool U8>
__device__ __forceinline__ Raw4<U8> load_raw4(const void* in, long long off) {
  Raw4<U8> r;
  if constexpr (U8) r.v = *reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(in) + off);
  else r.v = *reinterpret_cast<const float4*>(static_cast<const float*>(in) + off);
  return r;
}
// A/B switch for experiments: -DXT_GUARDED_LOADS turns the clamped unconditional loads back into guarded ones
template <bool U8>
__device__ __forceinline__ Raw4<U8> load_raw4_ok(const void* in, long long off, bool ok) {
#ifdef XT_GUARDED_LOADS
  Raw4<U8> r;
  if constexpr (U8) r.v = 0u; else r.v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) r = load_raw4<U8>(in, off);
  return r;
#else
  return load_raw4<U8>(in, ok ? off : 0ll);
#endif
}
__device__ __forceinline__ float4 load_f4_ok(const float* base, long long off, bool ok) {
#ifdef XT_GUARDED_LOADS
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) r = *reinterpret_cast<const float4*>(base + off);
  return r;
#else
  return *reinterpret_cast<const float4*>(base + (ok ? off : 0ll));
#endif
}

template <bool U8>
__device__ __forceinline__ float4 cook4(const Raw4<U8>& r, bool ok, float xs, float xb) {
  if constexpr (U8) {
    // v_cvt_f32_ubyteN + one fma per element
    return sel4(ok, make_float4(fmaf((float)(r.v & 0xffu), xs, xb), fmaf((float)((r.v >> 8) & 0xffu), xs, xb),
                                fmaf((float)((r.v >> 16) & 0xffu), xs, xb), fmaf((float)(r.v >> 24), xs, xb)));
  } else {
    return sel4(ok, r.v);
  }
}

// im2col row m -> sample index b and top-left input coordinate (no memory access, branch-free)
__device__ __forceinline__ void decode_coords(const Geom& g, int m, uint32_t* b, int* iy0, int* ix0) {
  const bool okm = m < g.M;
  const uint32_t mm = okm ? (uint32_t)m : 0u;
  const uint32_t bb = fdiv(mm, g.d_ohow);
  const uint32_t rem = mm - bb * (uint32_t)g.OHOW;
  const uint32_t oy = fdiv(rem, g.d_ow);
  const uint32_t ox = rem - oy * (uint32_t)g.OW;
  *b = bb;
  *iy0 = okm ? (int)oy * g.S - g.PT : -(1 << 28);
  *ix0 = (int)ox * g.S - g.PL;
}
// decode N rows -> element offset of the (possibly out-of-image, when padded) top-left input element of the
// receptive field, relative to the input base.  All minibatch-gather index loads are issued back to back.
template <int N>
__device__ __forceinline__ void decode_rows(const Geom& g, const int (&m)[N], const int32_t* __restrict__ idx,
                                            long long (&rowbase)[N], int (&iy0)[N], int (&ix0)[N]) {
  uint32_t b[N];
#pragma unroll
  for (int i = 0; i < N; ++i) decode_coords(g, m[i], &b[i], &iy0[i], &ix0[i]);
  int32_t sidx[N];
#pragma unroll
  for (int i = 0; i < N; ++i) sidx[i] = idx ? idx[b[i]] : (int32_t)b[i];
#pragma unroll
  for (int i = 0; i < N; ++i)
    rowbase[i] = (long long)sidx[i] * g.HWC + (long long)((iy0[i] * g.W + ix0[i]) * g.C);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// 32 reduction steps of the tile product on the matrix cores
template <int TI, int TJ, int SA, int SB>
__device__ __forceinline__ void mma_tile(const float* As, const float* Bs, int a_off, int b_off,
                                         f32x16 (&acc)[TI][TJ], int lane) {
  const int kl = lane >> 5, il = lane & 31;
  // XT_EXP_WGRAD_KK (experiment builds only, `make alt ALTFLAGS=-DXT_EXP_WGRAD_KK=6`; results are WRONG): issue only that
  // many of the step's 16 fp32 MFMAs -- 6 x 64 cycles = the matrix-pipe time of a bf16x6 step (12 x 32) with every load, LDS
  // write and barrier of the step left as it is: an UPPER BOUND of what moving the weight gradients to the bf16 pipe can
  // return (tools/experiments/README.md, round 5)
#ifndef XT_EXP_WGRAD_KK
#define XT_EXP_WGRAD_KK 16
#endif
#pragma unroll
  for (int kk = 0; kk < XT_EXP_WGRAD_KK; ++kk) {        // (reads of the whole step hoisted in front of the MFMAs: slower with three
    float a[TI], b[TJ];                    //  co-resident waves per SIMD -- conv2/conv3/Dense bwd +1.1/+1.1/+1.3 us)
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) a[ti] = As[(kk * 2 + kl) * SA + a_off + ti * 32 + il];
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) b[tj] = Bs[(kk * 2 + kl) * SB + b_off + tj * 32 + il];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj)
        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
  }
}

// ------------------------------------------------------------------ forward
struct FwdArgs {
  Geom g;
  const void* in;
  const int32_t* idx;
  const float* w;
  const float* bias;
  float* y;        // ksplit==1: final output; else partial [ksplit][M][N]
  int ksplit, kchunk;
  int xcd_chunked; // block ids remapped so that every XCD owns a contiguous run of the (m tile, n tile, k slice) order
};

// PADDED: the receptive field can leave the image (TF SAME) -> per-element bounds checks; VALID convs and
// dense layers skip them.  Row base offsets are computed once per block; a k-step costs one (ky,kx,c)
// decode per thread plus one 64-bit add per row.
// KG = 2: the block has two 4-wave groups that split the reduction range in halves, each with its own LDS
// stages (same barriers), combined through LDS at the end: halves the dependent step chain of kernels that have
// too few blocks to fill the chip (<= 1 block per CU) without partial buffers or a finish launch.
//
// X6 ("bf16x6", fp32 layers only): both operands are split into three bf16 planes when they are written to LDS and
// every 16-deep chunk is six v_mfma_f32_32x32x16_bf16 (see igemm_dgrad4_body for the arithmetic).  A (reduction axis
// contiguous in memory) lands in operand order [plane][chunk][k half][row][8 bf16] -> one ds_read_b128 per operand.
// B = W[k][n] (n contiguous) stays in its NATURAL orientation [plane][k][n] -- the split writes are 8 bytes, no
// transposition through scalar writes -- and the operand (8 consecutive k of one column) is gathered by two
// ds_read_b64_tr_b16: in every 16-lane group lanes 4j..4j+3 supply the address of 16 columns of row j and lane c
// receives column c of the four rows (semantics measured with tools/tr_probe.hip).  Row stride 160 B: the four rows of
// a read fall into disjoint bank ranges.
template <int BI, int BJ>
struct X6Lay {
  static constexpr int SlotA = BI * 16 + 32;            // bytes per (plane, chunk, k half) slot of BI rows
  static constexpr int RowB = BJ * 2 + 32, PlaneB = 32 * RowB;
  static constexpr int Stage = 12 * SlotA + 3 * PlaneB; // bytes per LDS stage
};
typedef short i16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_read_tr16x2(const uint8_t* p, int second_off) {
  union { i16x4 h[2]; bf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p + second_off));
  return u.v;
}

// NST > 0 (round 3): the block's reduction has at most NST steps per wave group and ALL of their operand loads are
// issued before the first one is consumed (NST register stages, straight-line code, exact s_waitcnt counts).  A step
// of the bf16x6 form is ~0.25 us of LDS + MFMA work, a global round trip under load 2-3 us: with two steps in flight
// the loop ran at the memory latency (timeline: 4.7 us for the 8 steps of PpoCnn's conv2, 14.8 us launch).  Steps
// past a group's range load clamped addresses and contribute zeros.
template <int BI, int BJ, int WI, int WJ, bool U8, bool PADDED, int KG, bool X6 = false, int NST = 0>
__global__ __launch_bounds__(256 * KG) void igemm_fwd_kernel(const FwdArgs p) {
  constexpr int TI = BI / (32 * WI), TJ = BJ / (32 * WJ);
  constexpr int SA = BI + 1, SB = BJ;
  constexpr int NA = BI / 32;            // 4-element fetches per thread for A
  constexpr int CPRB = BJ / 4;           // float4 groups per B row
  constexpr int RPB = 256 / CPRB;        // B rows per pass
  constexpr int NB = 32 / RPB;
  static_assert(!X6 || !U8, "igemm_fwd: the bf16x6 form is for fp32 inputs");
  constexpr int kX6SlotA = X6Lay<BI, BJ>::SlotA, kX6RowB = X6Lay<BI, BJ>::RowB, kX6PlaneB = X6Lay<BI, BJ>::PlaneB;
  constexpr int BUF = X6 ? X6Lay<BI, BJ>::Stage / 4 : 32 * SA + 32 * SB; // one LDS stage (A then B); two stages, one barrier per step
  // KG = 4 (round 3): four 4-wave groups, each a quarter of the reduction range with ONE LDS stage (two barriers per
  // step): a step costs ~1 us whatever it contains (ablations), so the lever is the number of steps a block
  // walks in sequence -- 16 waves per CU also give the phases of different groups something to overlap with.
  constexpr int NSTG = (KG == 4) ? 1 : 2;
  __shared__ __attribute__((aligned(16))) float smem_all[NSTG * BUF * KG];
  const Geom& g = p.g;
  const int grp = (KG == 1) ? 0 : (int)(threadIdx.x >> 8);
  float* smem = smem_all + grp * NSTG * BUF;
  const int t = threadIdx.x & 255, lane = t & 63, wave = t >> 6;
  // Blocks are dispatched round robin over the 8 XCDs in linear order (x fastest).  With more than one N tile or k
  // split, neighbours in that order share operand slices (the M tiles of one (n tile, k slice) stream the same weight
  // columns, the N tiles of one (m tile, k slice) the same input rows) and land on DIFFERENT L2s: ImpalaCnnOpt's
  // 11x11 layer at 128 frames fetched 21 MB for 6 MB of operands (round 2 PMC).  xcd_chunk gives every XCD a
  // contiguous run of the linear order instead.
  int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
  if (p.xcd_chunked) {
    const uint32_t nb = gridDim.x * gridDim.y * gridDim.z;
    uint32_t lin = xcd_chunk(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nb);
    bxi = (int)(lin % gridDim.x); lin /= gridDim.x;
    byi = (int)(lin % gridDim.y); bzi = (int)(lin / gridDim.y);
  }
  const int i0 = bxi * BI, j0 = byi * BJ;
  const int kbeg0 = bzi * p.kchunk;
  const int kend0 = min(g.K, kbeg0 + p.kchunk);
  const int nsteps_all = ((kend0 - kbeg0 + 31) / 32 + KG - 1) / KG;     // block-uniform loop length
  const int kbeg = kbeg0 + grp * nsteps_all * 32;
  const int kend = min(kend0, kbeg + nsteps_all * 32);
  XT_TL(0);
  XT_TL_ROLE(10);

  const int c4 = t & 7, r0 = t >> 3;
  long long rowbase[NA];
  int iy0[NA], ix0[NA];
  uint32_t rowok = 0;
  {
    int mrow[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) { mrow[i] = i0 + r0 + 32 * i; rowok |= (mrow[i] < g.M ? 1u : 0u) << i; }
    decode_rows<NA>(g, mrow, p.idx, rowbase, iy0, ix0);
  }
  const int cb = t % CPRB, rb = t / CPRB;
  const int nb = j0 + cb * 4;
  const bool nok = nb < g.N;

  struct Regs { Raw4<U8> a[NA]; float4 b[NB]; uint32_t ok; };
  auto fetch = [&](int k0, Regs& R) {
    const int k = k0 + c4 * 4;
    const bool kv = k < kend;
    const uint32_t ky = fdiv((uint32_t)k, g.d_kwc);
    const uint32_t r = (uint32_t)k - ky * (uint32_t)g.KWC;
    const uint32_t kx = fdiv(r, g.d_c);
    const int koff = ((int)ky * g.W + (int)kx) * g.C + (int)(r - kx * (uint32_t)g.C);
    R.ok = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      bool ok = kv && ((rowok >> i) & 1u);
      if (PADDED)
        ok = ok && ((unsigned)(iy0[i] + (int)ky) < (unsigned)g.H) && ((unsigned)(ix0[i] + (int)kx) < (unsigned)g.W);
      R.a[i] = load_raw4_ok<U8>(p.in, rowbase[i] + koff, ok);
      R.ok |= (ok ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int kk = k0 + rb + RPB * i;
      const bool okb = nok && kk < kend;
      R.b[i] = load_f4_ok(p.w, (long long)kk * g.N + nb, okb);
      R.ok |= (okb ? 1u : 0u) << (8 + i);
    }
  };
  auto stash = [&](const Regs& R, float* As, float* Bs) {
    if constexpr (X6) {
      uint8_t* Ap = reinterpret_cast<uint8_t*>(As);             // k = 4*c4 + e -> slot (c4 >> 1) = chunk*2 + k half
      uint8_t* Bp = Ap + 12 * kX6SlotA;
#pragma unroll
      for (int i = 0; i < NA; ++i)
        split3_store(Ap + (r0 + 32 * i) * 16 + (c4 >> 1) * kX6SlotA + (c4 & 1) * 8, 4 * kX6SlotA,
                     cook4<U8>(R.a[i], (R.ok >> i) & 1u, g.xs, g.xb));
#pragma unroll
      for (int i = 0; i < NB; ++i)
        split3_store(Bp + (rb + RPB * i) * kX6RowB + cb * 8, kX6PlaneB, sel4((R.ok >> (8 + i)) & 1u, R.b[i]));
      return;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = r0 + 32 * i;
      const float4 v = cook4<U8>(R.a[i], (R.ok >> i) & 1u, g.xs, g.xb);
      As[(c4 * 4 + 0) * SA + r] = v.x;
      As[(c4 * 4 + 1) * SA + r] = v.y;
      As[(c4 * 4 + 2) * SA + r] = v.z;
      As[(c4 * 4 + 3) * SA + r] = v.w;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      *reinterpret_cast<float4*>(&Bs[(rb + RPB * i) * SB + cb * 4]) = sel4((R.ok >> (8 + i)) & 1u, R.b[i]);
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

  const int wi = wave / WJ, wj = wave % WJ;
  const int nsteps = kend > kbeg ? (kend - kbeg + 31) / 32 : 0;      // this group's steps (<= nsteps_all)
  auto mma = [&](const float* stage) {
#if defined(XT_ABL) && XT_ABL == 2     // ablation: no operand reads, no MFMAs (timing probe)
    return;
#endif
    if constexpr (X6) {
      const uint8_t* Ap = reinterpret_cast<const uint8_t*>(stage);
      const uint8_t* Bp = Ap + 12 * kX6SlotA;
      const int kl = lane >> 5, il = lane & 31, c16 = lane & 15;
      const uint8_t* bq = Bp + (8 * kl + (c16 >> 2)) * kX6RowB + (wj * TJ * 32 + ((lane >> 4) & 1) * 16 + 4 * (c16 & 3)) * 2;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        bf16x8 a[TI][3], b[TJ][3];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            a[ti][pl] = *reinterpret_cast<const bf16x8*>(Ap + (pl * 4 + ch * 2 + kl) * kX6SlotA + ((wi * TI + ti) * 32 + il) * 16);
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            b[tj][pl] = lds_read_tr16x2(bq + pl * kX6PlaneB + ch * 16 * kX6RowB + tj * 64, 4 * kX6RowB);
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj) acc[ti][tj] = mfma_bf16x6(a[ti], b[tj], acc[ti][tj]);
      }
    } else {
      mma_tile<TI, TJ, SA, SB>(stage, stage + 32 * SA, wi * TI * 32, wj * TJ * 32, acc, lane);
    }
  };
  if constexpr (NST > 0) {
    Regs R[NST];
#pragma unroll
    for (int d = 0; d < NST; ++d) fetch(kbeg + 32 * d, R[d]);
    XT_TL(1);
#pragma unroll
    for (int d = 0; d < NST; ++d) {
      float* stage = smem + (d & 1) * BUF;
      stash(R[d], stage, stage + 32 * SA);
      __syncthreads();
      if (d == 0) XT_TL(2);
      mma(stage);
    }
  } else if constexpr (KG == 4) {
    Regs R0, R1;
    if (nsteps > 0) fetch(kbeg, R0);
    if (nsteps > 1) fetch(kbeg + 32, R1);
    XT_TL(1);
    for (int s = 0; s < nsteps_all; s += 2) {
      if (s < nsteps) stash(R0, smem, smem + 32 * SA);
      __syncthreads();
      if (s == 0) XT_TL(2);
      if (s + 2 < nsteps) fetch(kbeg + (s + 2) * 32, R0);
      if (s < nsteps) mma(smem);
      __syncthreads();
      if (s + 1 < nsteps_all) {
        if (s + 1 < nsteps) stash(R1, smem, smem + 32 * SA);
        __syncthreads();
        if (s + 3 < nsteps) fetch(kbeg + (s + 3) * 32, R1);
        if (s + 1 < nsteps) mma(smem);
        __syncthreads();
      }
    }
  } else {
  Regs R0, R1;
  if (nsteps > 0) fetch(kbeg, R0);
  if (nsteps > 1) fetch(kbeg + 32, R1);
  XT_TL(1);
  for (int s = 0; s < nsteps_all; s += 2) {
    if (s < nsteps) stash(R0, smem, smem + 32 * SA);
    __syncthreads();
    if (s == 0) XT_TL(2);
    if (s + 2 < nsteps) fetch(kbeg + (s + 2) * 32, R0);
    if (s < nsteps) mma(smem);
    if (s + 1 < nsteps_all) {
      if (s + 1 < nsteps) stash(R1, smem + BUF, smem + BUF + 32 * SA);
      __syncthreads();
      if (s + 3 < nsteps) fetch(kbeg + (s + 3) * 32, R1);
      if (s + 1 < nsteps) mma(smem + BUF);
    }
  }
  }
  if constexpr (KG >= 2) {
    // combine the groups' accumulators AND transpose: both groups park their tiles in LDS (every stage buffer
    // is idle after the barrier) as [group][row][BJ + 4]; then all 512 threads sum the two copies and store 16
    // bytes each (a row of the tile is contiguous in y).  The dword form (lanes = columns, two 128-byte rows per
    // instruction, group 1 idle) spent 2.5 us of a 10.6 us block in the store issue.
    __syncthreads();
    XT_TL(3);
    constexpr int RS = BJ + 4;
    static_assert(KG * BI * RS <= NSTG * BUF * KG, "igemm_fwd: combine buffer does not fit the stage buffers");
    float* red = smem_all;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wi * TI + ti) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          red[(grp * BI + row) * RS + (wj * TJ + tj) * 32 + (lane & 31)] = acc[ti][tj][r];
        }
    __syncthreads();
    const bool fin = (p.ksplit == 1);
    float* outp = fin ? p.y : p.y + (size_t)bzi * (size_t)g.M * g.N;
    for (int e = (int)threadIdx.x; e < BI * BJ / 4; e += 256 * KG) {
      const int row = e / (BJ / 4), c4 = (e - row * (BJ / 4)) * 4;
      float4 a = *reinterpret_cast<const float4*>(&red[row * RS + c4]);
#pragma unroll
      for (int q = 1; q < KG; ++q) {            // groups in order: a fixed summation order
        const float4 b = *reinterpret_cast<const float4*>(&red[(q * BI + row) * RS + c4]);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      float v[4] = {a.x, a.y, a.z, a.w};
      const int m = i0 + row, n = j0 + c4;
      if (m >= g.M) continue;
      if (fin) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = act_apply(v[q] + (n + q < g.N ? p.bias[n + q] : 0.f), g.act);
      }
      float* dst = outp + (size_t)m * g.N + n;
      if (n + 3 < g.N && (g.N & 3) == 0) store4_wt(outp, (size_t)m * g.N + n, make_float4(v[0], v[1], v[2], v[3]));
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (n + q < g.N) dst[q] = v[q];
      }
    }
    XT_TL(4);
    XT_TL_DRAIN(5);
    return;
  }

  XT_TL(3);
  const bool final_out = (p.ksplit == 1);
  float* out = final_out ? p.y : p.y + (size_t)bzi * (size_t)g.M * g.N;
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const int n = j0 + (wj * TJ + tj) * 32 + (lane & 31);
      const float bv = (final_out && n < g.N) ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = i0 + (wi * TI + ti) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < g.M && n < g.N) {
          float v = acc[ti][tj][r];
          if (final_out) v = act_apply(v + bv, g.act);
          out[(size_t)m * g.N + n] = v;
        }
      }
    }
  XT_TL(4);
  XT_TL_DRAIN(5);
}

// y = act(sum_z partial[z] + bias)
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                            float* __restrict__ y, int MN, int N, int ksplit, int act) {
  const int e4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e4 >= MN) return;
  float4 s = *reinterpret_cast<const float4*>(partial + e4);
  for (int z = 1; z < ksplit; ++z) {
    const float4 q = *reinterpret_cast<const float4*>(partial + (size_t)z * MN + e4);
    s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
  }
  const int n = e4 % N;
  const float4 b = *reinterpret_cast<const float4*>(bias + n);
  s.x = act_apply(s.x + b.x, act); s.y = act_apply(s.y + b.y, act);
  s.z = act_apply(s.z + b.z, act); s.w = act_apply(s.w + b.w, act);
  *reinterpret_cast<float4*>(y + e4) = s;
}

// ------------------------------------------------------------------ wgrad
struct WgradArgs {
  Geom g;
  const void* in;
  const int32_t* idx;
  const float* dy;
  float* out;      // [msplit][(K+1)*N]
  int msplit, mchunk;
  FastDiv d_rowq;  // wgrad_rows_body: divide by the float4 count of an input row
  int sq_gx;       // k tiles per n tile (row length of sq_out)
  float* sq_out;   // (msplit == 1 only, may be null) [tiles] sum of squares of each block's share of the gradient: the
                   // partials of the global norm, so that the reduction launch does not read the tensor again
};

// XT_EXP_WGRAD_X6 (experiment builds: `make alt ALTFLAGS=-DXT_EXP_WGRAD_X6=1`): the LDS-tiled fp32 weight gradients on the
// bf16 matrix cores with a CONSUMER-side split -- both operands split into three bf16 planes when they are written to LDS
// (natural [m][k] / [m][n] orientation, gathered by ds_read_b64_tr_b16 like the forward's weight operand), six
// v_mfma_f32_32x32x16_bf16 per 16 rows.  Round 2 measured this idea at < 0.3 us; round 5 measured the CEILING of the
// matrix-pipe relief at -8.5 us per step (XT_EXP_WGRAD_KK) and re-measured it with the round-3/4 loops -- numbers in
// tools/experiments/README.md.
#ifdef XT_EXP_WGRAD_X6
constexpr bool kWgradX6 = true;
#else
constexpr bool kWgradX6 = false;
#endif
constexpr int kRowTab = 1024;                       // rows decoded at once into the LDS row table
constexpr long long kRowInvalid = -(1ll << 62);

// Each thread owns a fixed group of 4 k's (its (ky,kx,c) offset is computed once); the rows of the block's
// m-range are decoded once into an LDS table (base offset [+ coordinates when PADDED]), so a reduction step
// costs one ds_read + one 64-bit add per row.
template <int BI, int BJ>
struct WgX6Lay {          // bf16x6 stage: [3 planes][32 rows][BI * 2 + 32 bytes] then [3][32][BJ * 2 + 32]
  static constexpr int RowA = BI * 2 + 32, RowB = BJ * 2 + 32, PlaneA = 32 * RowA, PlaneB = 32 * RowB;
  static constexpr int Stage = 3 * (PlaneA + PlaneB);      // bytes
};
template <int BI, int BJ, bool PADDED, int WX6 = 0>       // WX6: 0 fp32, 1 bf16x6 with two LDS stages, 2 with ONE stage
constexpr int wgrad_smem_floats() {
  return (WX6 == 2 ? 1 : 2) * ((kWgradX6 || WX6) ? WgX6Lay<BI, BJ>::Stage / 4 : (32 * BI + 32 * BJ)) + 2 * kRowTab +
         (PADDED ? kRowTab : 0);
}

// PF4 (round 3): four register stages instead of two -- a step is 16 MFMAs (0.43 us), a global round trip under load
// 2-3 us: with two steps in flight the m loop ran at ~1 us per step (timeline).  Needs the 256-VGPR budget of a kernel
// instance limited to two workgroups per CU.
// WX6 (round 5): the weight gradient on the bf16 matrix cores -- both operands split into three bf16 planes when they
// are written to LDS (natural [m][k] / [m][n] orientation, row strides BI * 2 + 32 / BJ * 2 + 32 bytes), gathered by
// ds_read_b64_tr_b16 like the forward's weight operand, six v_mfma_f32_32x32x16_bf16 per 16 rows; the bias column sums add
// the three planes back up (exactly the fp32 values).  Measured layer by layer (tools/experiments/README.md, round 5): pays
// for the Dense layer's backward (18.04 -> 16.90 us: ten short steps, the co-resident input-gradient blocks get the pipe),
// LOSES for the conv layers (more elements to split per step; conv3 also drops from three to two workgroups per CU) -- so
// only the Dense instance of the fused backward selects it (xt_tuning.dense_wgrad_x6).
template <int BI, int BJ, int WI, int WJ, bool U8, bool PADDED, bool PF4 = false, int WX6 = 0>
__device__ __forceinline__ void igemm_wgrad_body(const WgradArgs& p, const int bx, const int by, const int bz,
                                                 float* smem) {
  constexpr int TI = BI / (32 * WI), TJ = BJ / (32 * WJ);
  constexpr int SA = BI, SB = BJ;
  constexpr int CPRA = BI / 4, RPA = 256 / CPRA, NA = 32 / RPA;
  constexpr int CPRB = BJ / 4, RPB = 256 / CPRB, NB = 32 / RPB;
  constexpr int RG = 256 / BJ;           // row groups for the bias column sums
  constexpr bool X6W = (kWgradX6 || WX6 != 0) && !U8;
  constexpr bool X6S = X6W && WX6 == 2;      // one LDS stage (half the LDS: a third workgroup per CU), two barriers per step
  using XL = WgX6Lay<BI, BJ>;
  constexpr int BUF = X6W ? XL::Stage / 4 : 32 * SA + 32 * SB;
  constexpr int BUF2 = X6S ? 0 : BUF;         // offset of the second stage (none in the one-stage form)
  long long* rowtab = reinterpret_cast<long long*>(smem + (X6S ? 1 : 2) * BUF);
  int* rowxy = reinterpret_cast<int*>(smem + (X6S ? 1 : 2) * BUF + 2 * kRowTab);
  const Geom& g = p.g;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i0 = bx * BI, j0 = by * BJ;     // i = k, j = n
  const int mbeg = bz * p.mchunk;
  const int mend = min(g.M, mbeg + p.mchunk);

  // this thread's fixed k group
  const int ca = t % CPRA, rowa = t / CPRA;
  const int k = i0 + ca * 4;
  const bool kok = k < g.K;
  const uint32_t ky = fdiv((uint32_t)k, g.d_kwc);
  const uint32_t rr = (uint32_t)k - ky * (uint32_t)g.KWC;
  const uint32_t kx = fdiv(rr, g.d_c);
  const int koff = ((int)ky * g.W + (int)kx) * g.C + (int)(rr - kx * (uint32_t)g.C);
  const int cb = t % CPRB, rowb = t / CPRB;
  const int nb = j0 + cb * 4;
  const bool nok = nb < g.N;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

  const int wi = wave / WJ, wj = wave % WJ;
  const bool do_bias = (bx == 0);
  const int bcol = t % BJ, brg = t / BJ;
  float bsum = 0.f;
  XT_TL(0);
  XT_TL_ROLE(20);

  for (int sub = mbeg; sub < mend; sub += kRowTab) {
    const int sub_end = min(mend, sub + kRowTab);
    const int nsteps = (sub_end - sub + 31) / 32;
    // "lean" loads (fp32 input, no row gather): see below
    constexpr bool LEAN_T = !U8;
    const bool lean = LEAN_T && p.idx == nullptr && (long long)g.B * g.HWC * 4 < (1ll << 30) &&
                      (long long)g.M * g.N * 4 < (1ll << 30);
    uint32_t* tab32 = reinterpret_cast<uint32_t*>(rowtab);
    // ---- decode the rows of this sub-range once
    for (int r = t; r < nsteps * 32; r += 256) {
      int mr[1] = {sub + r < sub_end ? sub + r : g.M};
      long long base[1];
      int iy0[1], ix0[1];
      decode_rows<1>(g, mr, p.idx, base, iy0, ix0);
      if (lean) {      // 32-bit byte offsets, the NA rows a thread loads in one step side by side: one LDS read per step
        const int w = r & 31, slot = (r & ~31) + (w % RPA) * NA + w / RPA;
        const bool rv = sub + r < sub_end;
        // (padded: the offset of an out-of-image top-left corner wraps, every tap is range-checked by its coordinates;
        // an invalid row gets coordinates that fail every check)
        tab32[slot] = (rv || PADDED) ? (uint32_t)((long long)base[0] * 4) : kOob;
        if (PADDED) rowxy[slot] = rv ? ((iy0[0] << 16) | (ix0[0] & 0xffff)) : (int)0xC000C000;
        continue;
      }
      rowtab[r] = (sub + r < sub_end) ? base[0] : kRowInvalid;
      if (PADDED) rowxy[r] = (iy0[0] << 16) | (ix0[0] & 0xffff);
    }
    __syncthreads();

    struct Regs { Raw4<U8> a[NA]; float4 b[NB]; uint32_t ok; };
    auto fetch = [&](int s, Regs& R) {
      R.ok = 0;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int r = s * 32 + rowa + RPA * i;
        const long long rbse = rowtab[r];
        bool ok = kok && (rbse > kRowInvalid);
        if (PADDED) {
          const int xy = rowxy[r];
          ok = ok && ((unsigned)((xy >> 16) + (int)ky) < (unsigned)g.H) &&
               ((unsigned)((int)(short)(xy & 0xffff) + (int)kx) < (unsigned)g.W);
        }
        R.a[i] = load_raw4_ok<U8>(p.in, rbse + koff, ok);
        R.ok |= (ok ? 1u : 0u) << i;
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int m = sub + s * 32 + rowb + RPB * i;
        const bool okb = nok && m < sub_end;
        R.b[i] = load_f4_ok(p.dy, (long long)m * g.N + nb, okb);
        R.ok |= (okb ? 1u : 0u) << (8 + i);
      }
    };
    auto stash = [&](const Regs& R, float* As, float* Bs) {
      if constexpr (X6W) {
        uint8_t* Ap = reinterpret_cast<uint8_t*>(As);
        uint8_t* Bp = Ap + 3 * XL::PlaneA;
#pragma unroll
        for (int i = 0; i < NA; ++i)
          split3_store(Ap + (rowa + RPA * i) * XL::RowA + ca * 8, XL::PlaneA, cook4<U8>(R.a[i], (R.ok >> i) & 1u, g.xs, g.xb));
#pragma unroll
        for (int i = 0; i < NB; ++i)
          split3_store(Bp + (rowb + RPB * i) * XL::RowB + cb * 8, XL::PlaneB, sel4((R.ok >> (8 + i)) & 1u, R.b[i]));
        return;
      }
#pragma unroll
      for (int i = 0; i < NA; ++i)
        *reinterpret_cast<float4*>(&As[(rowa + RPA * i) * SA + ca * 4]) =
            cook4<U8>(R.a[i], (R.ok >> i) & 1u, g.xs, g.xb);
#pragma unroll
      for (int i = 0; i < NB; ++i)
        *reinterpret_cast<float4*>(&Bs[(rowb + RPB * i) * SB + cb * 4]) = sel4((R.ok >> (8 + i)) & 1u, R.b[i]);
    };
    auto colsum = [&](const float* stage) {       // (stage base: A part first, then B)
      if (do_bias) {
        if constexpr (X6W) {      // the three planes add up to the fp32 value exactly: same sums, bit for bit
          const uint8_t* Bp = reinterpret_cast<const uint8_t*>(stage) + 3 * XL::PlaneA;
#pragma unroll
          for (int q = 0; q < 32 / RG; ++q) {
            const uint8_t* e = Bp + (brg + RG * q) * XL::RowB + bcol * 2;
            const float h = __uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(e) << 16);
            const float m2 = __uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(e + XL::PlaneB) << 16);
            const float l = __uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(e + 2 * XL::PlaneB) << 16);
            bsum += (h + m2) + l;
          }
          return;
        }
        const float* Bs = stage + 32 * SA;
#pragma unroll
        for (int q = 0; q < 32 / RG; ++q) bsum += Bs[(brg + RG * q) * SB + bcol];
      }
    };
    // the 32-row step on the matrix cores: fp32 MFMAs, or (X6W) two 16-row chunks of six bf16 MFMAs with both operands
    // gathered from their natural-orientation planes by ds_read_b64_tr_b16
    auto mma_step = [&](const float* stage) {
      if constexpr (X6W) {
        const uint8_t* Ap = reinterpret_cast<const uint8_t*>(stage);
        const uint8_t* Bp = Ap + 3 * XL::PlaneA;
        const int kl = lane >> 5, c16 = lane & 15;
        const int colq = ((lane >> 4) & 1) * 16 + 4 * (c16 & 3);
        const uint8_t* aq = Ap + (8 * kl + (c16 >> 2)) * XL::RowA + (wi * TI * 32 + colq) * 2;
        const uint8_t* bq = Bp + (8 * kl + (c16 >> 2)) * XL::RowB + (wj * TJ * 32 + colq) * 2;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          bf16x8 a[TI][3], b[TJ][3];
#pragma unroll
          for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              a[ti][pl] = lds_read_tr16x2(aq + pl * XL::PlaneA + ch * 16 * XL::RowA + ti * 64, 4 * XL::RowA);
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              b[tj][pl] = lds_read_tr16x2(bq + pl * XL::PlaneB + ch * 16 * XL::RowB + tj * 64, 4 * XL::RowB);
#pragma unroll
          for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj) acc[ti][tj] = mfma_bf16x6(a[ti], b[tj], acc[ti][tj]);
        }
      } else {
        mma_tile<TI, TJ, SA, SB>(stage, stage + 32 * SA, wi * TI * 32, wj * TJ * 32, acc, lane);
      }
    };

    // A wave does not issue its other instructions in the shadow of its own MFMA chain (tools/mfma_probe.hip: 64 cycles per
    // dependent v_mfma_f32_32x32x2_f32 bare, 113 with ten independent VALU operations behind each), so with one such wave
    // per SIMD every instruction of the step that is not an MFMA costs ~5 cycles of the step.  The lean form of the
    // four-stage loop removes the ones that only exist for addressing and masking:
    //  * buffer loads with 32-bit byte offsets; a row outside the range carries an offset beyond the buffer and reads as
    //    zeros (hardware range check): no validity bits, no 64-bit address arithmetic, one add per load;
    //  * the LDS writes store the load registers as they are (no zero-selects);
    //  * one LDS read per step fetches all NA row offsets of the thread.
    if constexpr (LEAN_T) if (lean) {
      const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.in, (uint32_t)g.B * (uint32_t)g.HWC * 4u);
      const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.dy, (uint32_t)g.M * (uint32_t)g.N * 4u);
      const uint32_t koff4 = (kok || PADDED) ? (uint32_t)koff * 4u : 0x40000000u;   // (any row offset + 2^30 is out of range)
      struct LRegs { float4 a[NA]; float4 b[NB]; };
      auto lfetch = [&](int sf, LRegs& Rf) {
        uint32_t ro[NA];
        int rxy[NA];
        if constexpr (NA == 4) {
          const uint4 q = *reinterpret_cast<const uint4*>(&tab32[sf * 32 + rowa * 4]);
          ro[0] = q.x; ro[1] = q.y; ro[2] = q.z; ro[3] = q.w;
          if constexpr (PADDED) {
            const int4 c = *reinterpret_cast<const int4*>(&rowxy[sf * 32 + rowa * 4]);
            rxy[0] = c.x; rxy[1] = c.y; rxy[2] = c.z; rxy[3] = c.w;
          }
        } else {
#pragma unroll
          for (int i = 0; i < NA; ++i) {
            ro[i] = tab32[sf * 32 + rowa * NA + i];
            if constexpr (PADDED) rxy[i] = rowxy[sf * 32 + rowa * NA + i];
          }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          uint32_t vo = ro[i] + koff4;
          if constexpr (PADDED) {
            const bool ok = kok && ((unsigned)((rxy[i] >> 16) + (int)ky) < (unsigned)g.H) &&
                            ((unsigned)((int)(short)(rxy[i] & 0xffff) + (int)kx) < (unsigned)g.W);
            vo = ok ? vo : kOob;
          }
          Rf.a[i] = buf_load4(rs_a, vo, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int m = sub + sf * 32 + rowb + RPB * i;
          Rf.b[i] = buf_load4(rs_b, (nok && m < sub_end) ? (uint32_t)(m * g.N + nb) * 4u : kOob, 0);
        }
      };
      auto lstash = [&](const LRegs& Rs, float* As, float* Bs) {
        if constexpr (X6W) {
          uint8_t* Ap = reinterpret_cast<uint8_t*>(As);
          uint8_t* Bp = Ap + 3 * XL::PlaneA;
#pragma unroll
          for (int i = 0; i < NA; ++i) split3_store(Ap + (rowa + RPA * i) * XL::RowA + ca * 8, XL::PlaneA, Rs.a[i]);
#pragma unroll
          for (int i = 0; i < NB; ++i) split3_store(Bp + (rowb + RPB * i) * XL::RowB + cb * 8, XL::PlaneB, Rs.b[i]);
          return;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<float4*>(&As[(rowa + RPA * i) * SA + ca * 4]) = Rs.a[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(&Bs[(rowb + RPB * i) * SB + cb * 4]) = Rs.b[i];
      };
      if constexpr (PF4) {
        LRegs R[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) lfetch(u < nsteps ? u : nsteps - 1, R[u]);
        if (sub == mbeg) XT_TL(1);
        for (int s = 0; s < nsteps; s += 4) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (s + u < nsteps) {                                                   // block-uniform
              float* stage = smem + (u & 1) * BUF2;
              lstash(R[u], stage, stage + 32 * SA);
              __syncthreads();
              if (s == 0 && u == 0 && sub == mbeg) XT_TL(2);
              if (s + u + 4 < nsteps) lfetch(s + u + 4, R[u]);
              colsum(stage);
              mma_step(stage);
              if constexpr (X6S) __syncthreads();
            }
          }
        }
      } else {
        LRegs R0, R1;
        if (nsteps > 0) lfetch(0, R0);
        if (nsteps > 1) lfetch(1, R1);
        if (sub == mbeg) XT_TL(1);
        for (int s = 0; s < nsteps; s += 2) {
          lstash(R0, smem, smem + 32 * SA);
          __syncthreads();
          if (s == 0 && sub == mbeg) XT_TL(2);
          if (s + 2 < nsteps) lfetch(s + 2, R0);
          colsum(smem);
          mma_step(smem);
          if constexpr (X6S) __syncthreads();
          if (s + 1 < nsteps) {
            lstash(R1, smem + BUF2, smem + BUF2 + 32 * SA);
            __syncthreads();
            if (s + 3 < nsteps) lfetch(s + 3, R1);
            colsum(smem + BUF2);
            mma_step(smem + BUF2);
            if constexpr (X6S) __syncthreads();
          }
        }
      }
      __syncthreads();
      continue;
    }
    if constexpr (PF4) {
      Regs R[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) fetch(u < nsteps ? u : nsteps - 1, R[u]);      // (clamped: steps past the range repeat the last)
      if (sub == mbeg) XT_TL(1);
      for (int s = 0; s < nsteps; s += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (s + u < nsteps) {                                                   // block-uniform
            float* stage = smem + (u & 1) * BUF2;
            stash(R[u], stage, stage + 32 * SA);
            __syncthreads();
            if (s == 0 && u == 0 && sub == mbeg) XT_TL(2);
            if (s + u + 4 < nsteps) fetch(s + u + 4, R[u]);
            colsum(stage);
            mma_step(stage);
            if constexpr (X6S) __syncthreads();
          }
        }
      }
      __syncthreads();
      continue;
    }
    Regs R0, R1;
    if (nsteps > 0) fetch(0, R0);
    if (nsteps > 1) fetch(1, R1);
    if (sub == mbeg) XT_TL(1);
    for (int s = 0; s < nsteps; s += 2) {
      stash(R0, smem, smem + 32 * SA);
      __syncthreads();
      if (s == 0 && sub == mbeg) XT_TL(2);
      if (s + 2 < nsteps) fetch(s + 2, R0);
      colsum(smem);
      mma_step(smem);
      if constexpr (X6S) __syncthreads();
      if (s + 1 < nsteps) {
        stash(R1, smem + BUF2, smem + BUF2 + 32 * SA);
        __syncthreads();
        if (s + 3 < nsteps) fetch(s + 3, R1);
        colsum(smem + BUF2);
        mma_step(smem + BUF2);
        if constexpr (X6S) __syncthreads();
      }
    }
    __syncthreads();   // row table and LDS stages are reused by the next sub-range
  }

  XT_TL(3);
  float* out = p.out + (size_t)bz * ((size_t)(g.K + 1) * g.N);
  float sq = 0.f;
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const int n = j0 + (wj * TJ + tj) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kr = i0 + (wi * TI + ti) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (kr < g.K && n < g.N) { store1_wt(out, (size_t)kr * g.N + n, acc[ti][tj][r]); sq += acc[ti][tj][r] * acc[ti][tj][r]; }
      }
    }
  if (do_bias) {
    smem[brg * BJ + bcol] = bsum;      // safe: the loop ended with a barrier
    __syncthreads();
    if (t < BJ) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < RG; ++q) sum += smem[q * BJ + t];
      const int n = j0 + t;
      if (n < g.N) { out[(size_t)g.K * g.N + n] = sum; sq += sum * sum; }
    }
  }
  if (p.sq_out) {      // (single slab: `out` IS the gradient) this block's share of the squared global norm, fixed order
    sq = wave_sum(sq);
    __syncthreads();
    if (lane == 0) smem[wave] = sq;
    __syncthreads();
    if (t == 0) p.sq_out[by * p.sq_gx + bx] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
  }
  XT_TL(4);
  XT_TL_DRAIN(5);
}

template <int BI, int BJ, int WI, int WJ, bool U8, bool PADDED>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(const WgradArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[wgrad_smem_floats<BI, BJ, PADDED>()];
  igemm_wgrad_body<BI, BJ, WI, WJ, U8, PADDED>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// dst[e] = sum_z src[z][e]
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int count, int nslab) {
  const int e4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e4 >= count) return;
  float4 s = *reinterpret_cast<const float4*>(src + e4);
  for (int z = 1; z < nslab; ++z) {
    const float4 q = *reinterpret_cast<const float4*>(src + (size_t)z * count + e4);
    s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
  }
  *reinterpret_cast<float4*>(dst + e4) = s;
}

// ------------------------------------------------------------------ dgrad
constexpr int kMaxClasses = 16;      // stride-parity classes (S <= 4)
struct DgradArgs {
  Geom g;
  const float* dy;
  const float* w;
  const float* x;     // producer's post-activation output [B,H,W,C]
  float* dx;
  int act_prev;
  const uint32_t* xmask;   // optional (act_prev = relu, C = 32): sign mask of x, one word per pixel -- replaces the x reads
  FastDiv d_hw[kMaxClasses], d_w[kMaxClasses];   // per class: divide by HC*WC and by WC (row decode without idiv)
};

// X6 ("bf16x6"): both operands have the reduction axis contiguous in memory (dY rows along n, W rows along n), so both
// go to LDS in MFMA operand order [plane][chunk][k half][row][8 bf16] with 8-byte split writes and are read back with
// one ds_read_b128 per operand; six v_mfma_f32_32x32x16_bf16 per 16-deep chunk (see igemm_dgrad4_body).
template <int BI, int BJ, bool X6 = false>
constexpr int dgrad_smem_floats() {
  return X6 ? 2 * (12 * (BI * 16 + 32) + 12 * (BJ * 16 + 32)) / 4 + BI : 2 * (32 * (BI + 1) + 32 * (BJ + 1)) + BI;
}

// NST > 0 (round 3): the class reduction has at most NST 32-deep steps and ALL of their operand loads are issued before
// the first is consumed (NST register stages; needs the 256-VGPR budget of a two-workgroups-per-CU instance).  With
// two steps in flight the 8-step loop of the Dense layer's input gradient ran at the memory latency, 1.2 us per step.
template <int BI, int BJ, int WI, int WJ, bool X6 = false, int NST = 0>
__device__ __forceinline__ void igemm_dgrad_body(const DgradArgs& p, const int bx, const int by, const int bz,
                                                 float* smem) {
  constexpr int TI = BI / (32 * WI), TJ = BJ / (32 * WJ);
  constexpr int SA = BI + 1, SB = BJ + 1;
  constexpr int NA = BI / 32, NB = BJ / 32;
  constexpr int SLA = BI * 16 + 32, SLB = BJ * 16 + 32;      // X6: bytes per (plane, chunk, k half) slot
  constexpr int BUF = X6 ? (12 * SLA + 12 * SLB) / 4 : 32 * SA + 32 * SB;
  int* rowOut = reinterpret_cast<int*>(smem + 2 * BUF);
  const Geom& g = p.g;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

  // stride-parity class of this block
  const int ry = bz / g.S, rx = bz % g.S;
  const int cy0 = ((ry - g.PT) % g.S + g.S) % g.S, cx0 = ((rx - g.PL) % g.S + g.S) % g.S;
  const int HC = cy0 < g.H ? (g.H - cy0 + g.S - 1) / g.S : 0;
  const int WC = cx0 < g.W ? (g.W - cx0 + g.S - 1) / g.S : 0;
  const int Mc = g.B * HC * WC;
  const int i0 = bx * BI, j0 = by * BJ;      // i = class pixel, j = input channel
  if (i0 >= Mc) return;
  const int JY = ry < g.KH ? (g.KH - ry + g.S - 1) / g.S : 0;
  const int JX = rx < g.KW ? (g.KW - rx + g.S - 1) / g.S : 0;
  const int Kc = JY * JX * g.N;
  const int qy0 = (cy0 + g.PT) / g.S, qx0 = (cx0 + g.PL) / g.S;
  XT_TL(0);
  XT_TL_ROLE(30);

  const FastDiv dhw = p.d_hw[bz], dw = p.d_w[bz];
  if (t < BI) {
    const int mc = i0 + t;
    int off = -1;
    if (mc < Mc) {
      const int b = (int)fdiv((uint32_t)mc, dhw), rem = mc - b * (HC * WC);
      const int ty = (int)fdiv((uint32_t)rem, dw), tx = rem - ty * WC;
      off = ((b * g.H + cy0 + g.S * ty) * g.W + cx0 + g.S * tx) * g.C;
    }
    rowOut[t] = off;
  }

  const int c4 = t & 7, r0 = t >> 3;
  int rowbase[NA], qy[NA], qx[NA];       // dY element offset of output pixel (qy,qx) of the row's sample
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int mc = i0 + r0 + 32 * i;
    if (mc < Mc) {
      const int b = (int)fdiv((uint32_t)mc, dhw), rem = mc - b * (HC * WC);
      const int ty = (int)fdiv((uint32_t)rem, dw), tx = rem - ty * WC;
      qy[i] = qy0 + ty; qx[i] = qx0 + tx;
      rowbase[i] = (b * g.OHOW + qy[i] * g.OW + qx[i]) * g.N;
    } else {
      rowbase[i] = 0; qy[i] = -(1 << 28); qx[i] = 0;
    }
  }
  int cN[NB];
  uint32_t cok = 0;
#pragma unroll
  for (int i = 0; i < NB; ++i) { const int c = j0 + r0 + 32 * i; cN[i] = c * g.N; cok |= (c < g.C ? 1u : 0u) << i; }

  struct Regs { float4 a[NA]; float4 b[NB]; uint32_t ok; };
  // (buffer loads: 32-bit offsets, taps / rows / channels outside the problem read as zeros through the range check, the
  // LDS writes take the registers as they are -- every instruction of the step that is not an MFMA costs the wave ~5
  // cycles, see igemm_wgrad_body)
  const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(p.dy, (uint32_t)g.M * (uint32_t)g.N * 4u);
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (uint32_t)g.K * (uint32_t)g.N * 4u);
  auto fetch = [&](int k0, Regs& R) {
    const int kk = k0 + c4 * 4;
    const bool kok = kk < Kc;
    R.ok = 0xffffffffu;
    const uint32_t tap = fdiv((uint32_t)kk, g.d_n);
    const int n = kk - (int)tap * g.N;
    const int jy = JX > 0 ? (int)tap / JX : 0, jx = (int)tap - jy * JX;
    const int tapoff = (jy * g.OW + jx) * g.N - n;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const bool ok = kok && ((unsigned)(qy[i] - jy) < (unsigned)g.OH) && ((unsigned)(qx[i] - jx) < (unsigned)g.OW);
      R.a[i] = buf_load4(rs_dy, ok ? (uint32_t)(rowbase[i] - tapoff) * 4u : kOob, 0);
    }
    const int wbase = ((ry + g.S * jy) * g.KW + rx + g.S * jx) * g.C * g.N + n;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const bool okb = kok && ((cok >> i) & 1u);
      R.b[i] = buf_load4(rs_w, okb ? (uint32_t)(wbase + cN[i]) * 4u : kOob, 0);
    }
  };
  auto stash = [&](const Regs& R, float* As, float* Bs) {
    if constexpr (X6) {
      uint8_t* Ap = reinterpret_cast<uint8_t*>(As);             // k = 4*c4 + e -> slot (c4 >> 1) = chunk*2 + k half
      uint8_t* Bp = Ap + 12 * SLA;
#pragma unroll
      for (int i = 0; i < NA; ++i)
        split3_store(Ap + (r0 + 32 * i) * 16 + (c4 >> 1) * SLA + (c4 & 1) * 8, 4 * SLA, R.a[i]);
#pragma unroll
      for (int i = 0; i < NB; ++i)
        split3_store(Bp + (r0 + 32 * i) * 16 + (c4 >> 1) * SLB + (c4 & 1) * 8, 4 * SLB, R.b[i]);
      return;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = r0 + 32 * i;
      const float4 v = R.a[i];
      As[(c4 * 4 + 0) * SA + r] = v.x;
      As[(c4 * 4 + 1) * SA + r] = v.y;
      As[(c4 * 4 + 2) * SA + r] = v.z;
      As[(c4 * 4 + 3) * SA + r] = v.w;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int r = r0 + 32 * i;
      const float4 v = R.b[i];
      Bs[(c4 * 4 + 0) * SB + r] = v.x;
      Bs[(c4 * 4 + 1) * SB + r] = v.y;
      Bs[(c4 * 4 + 2) * SB + r] = v.z;
      Bs[(c4 * 4 + 3) * SB + r] = v.w;
    }
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

  const int wi = wave / WJ, wj = wave % WJ;
  const int nsteps = (Kc + 31) / 32;
  Regs R0, R1;
  Regs RN[NST > 0 ? NST : 1];
  if constexpr (NST > 0) {
#pragma unroll
    for (int d = 0; d < NST; ++d) fetch(32 * d, RN[d]);      // (steps past Kc load a clamped address and contribute zeros)
  } else {
    if (nsteps > 0) fetch(0, R0);
    if (nsteps > 1) fetch(32, R1);
  }
  // prefetch the producer activations of this thread's output elements (clamped, unconditional): their latency
  // hides behind the reduction loop instead of being exposed in the epilogue (4.5 of 13.9 us per block before)
  __syncthreads();                 // rowOut
  float xv[TI][TJ][16];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const int c = j0 + (wj * TJ + tj) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int off = rowOut[(wi * TI + ti) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
        xv[ti][tj][r] = p.x[(off >= 0 && c < g.C) ? (size_t)off + c : (size_t)0];
      }
    }
  XT_TL(1);
  auto mma = [&](const float* stage) {
    if constexpr (X6) {
      const uint8_t* Ap = reinterpret_cast<const uint8_t*>(stage);
      const uint8_t* Bp = Ap + 12 * SLA;
      const int kl = lane >> 5, il = lane & 31;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        bf16x8 a[TI][3], b[TJ][3];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            a[ti][pl] = *reinterpret_cast<const bf16x8*>(Ap + (pl * 4 + ch * 2 + kl) * SLA + ((wi * TI + ti) * 32 + il) * 16);
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            b[tj][pl] = *reinterpret_cast<const bf16x8*>(Bp + (pl * 4 + ch * 2 + kl) * SLB + ((wj * TJ + tj) * 32 + il) * 16);
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj) acc[ti][tj] = mfma_bf16x6(a[ti], b[tj], acc[ti][tj]);
      }
    } else {
      mma_tile<TI, TJ, SA, SB>(stage, stage + 32 * SA, wi * TI * 32, wj * TJ * 32, acc, lane);
    }
  };
  if constexpr (NST > 0) {
#pragma unroll
    for (int d = 0; d < NST; ++d) {
      if (d < nsteps) {                          // block-uniform
        float* stage = smem + (d & 1) * BUF;
        stash(RN[d], stage, stage + 32 * SA);
        __syncthreads();
        if (d == 0) XT_TL(2);
        mma(stage);
      }
    }
  } else {
  for (int s = 0; s < nsteps; s += 2) {
    stash(R0, smem, smem + 32 * SA);
    __syncthreads();
    if (s == 0) XT_TL(2);
    if (s + 2 < nsteps) fetch((s + 2) * 32, R0);
    mma(smem);
    if (s + 1 < nsteps) {
      stash(R1, smem + BUF, smem + BUF + 32 * SA);
      __syncthreads();
      if (s + 3 < nsteps) fetch((s + 3) * 32, R1);
      mma(smem + BUF);
    }
  }
  }
  XT_TL(3);

  // epilogue: the producer activations (for act') were prefetched before the reduction loop (xv), so the
  // masked gradients are stored without a second exposed memory round trip.
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const int c = j0 + (wj * TJ + tj) * 32 + (lane & 31);
      const bool cok = c < g.C;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int off = rowOut[(wi * TI + ti) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
        if (off >= 0 && cok) store1_wt(p.dx, (size_t)off + c, acc[ti][tj][r] * act_grad(xv[ti][tj][r], p.act_prev));
      }
    }
  XT_TL(4);
  XT_TL_DRAIN(5);
}

// ------------------------------------------------------------------ dgrad, all stride-parity classes per block
// For a VALID stride-S conv whose kernel and input extents are multiples of S, the S*S parity classes of the
// same class position (b, ty, tx) gather the SAME dY pixels (ty - jy, tx - jx) and differ only in the weight
// taps (ky, kx) = (ry + S*jy, rx + S*jx).  One block therefore loads the dY tile of 128 positions ONCE per step and
// runs S*S = 4 accumulator tiles against 4 weight tiles: a quarter of the A traffic and of the blocks (250
// instead of 1000 for PpoCnn's 4x4/2 conv2 at B=320 -> the whole fused backward launch is co-resident in one
// round), four independent MFMA chains per wave, prologue/epilogue amortised over four times the math.
// Single LDS stage (two barriers per step): a step carries 64 MFMAs per wave, the barrier is noise, and the small
// footprint (33.5 KB) keeps three blocks per CU next to the weight-gradient blocks.
//
// SPLIT = true ("bf16x6"): fp32 x fp32 on the bf16 matrix cores.  Both operands are split into three bf16 planes
// (x = x1 + x2 + x3, truncation splits: 3 x 8 = 24 significant bits, i.e. all of fp32) when they are written to
// LDS -- once per element per block -- and each 16-deep chunk is accumulated as the six products
// x1*w1 + x1*w2 + x2*w1 + x1*w3 + x2*w2 + x3*w1 in fp32 (every bf16 x bf16 product is exact in fp32; the three
// dropped terms are below 2^-24 relative, the size of fp32's own rounding).  6 x v_mfma_f32_32x32x16_bf16 (32
// cycles each) replace 8 x v_mfma_f32_32x32x2_f32 (64 cycles each): 2.7x the matrix rate, which is what bounds
// this loop (73 % MFMA utilisation on its SIMDs, profiles/r01_timeline_late.txt).  LDS layout per operand:
// [plane][chunk][k half][row][8 bf16] = one ds_read_b128 per MFMA operand, slot stride padded by 32 B so that the
// 8-byte split writes of the eight k-quads of a row spread over all banks.
constexpr int kD4Classes = 4;
constexpr int kD4SlotA = 128 * 16 + 32, kD4SlotB = 32 * 16 + 32;     // bytes per (plane, chunk, k half) slot
template <bool SPLIT>
constexpr int dgrad4_smem_floats() {
  return SPLIT ? (12 * kD4SlotA + kD4Classes * 12 * kD4SlotB) / 4 + 128 : 32 * 129 + kD4Classes * 32 * 33 + 128;
}

template <bool SPLIT, bool PF4 = false>
__device__ __forceinline__ void igemm_dgrad4_body(const DgradArgs& p, const int bx, float* smem) {
  constexpr int BI = 128, SA = BI + 1, SB = 33, NA = 4;
  float* As = smem;
  float* Bs = smem + 32 * SA;                       // [class][32 k][SB]
  uint8_t* Ap = reinterpret_cast<uint8_t*>(smem);   // SPLIT: [12 slots][128 rows][16 B]
  uint8_t* Bp = Ap + 12 * kD4SlotA;                 //        [class][12 slots][32 cols][16 B]
  int* rowOut = SPLIT ? reinterpret_cast<int*>(Bp + kD4Classes * 12 * kD4SlotB)
                      : reinterpret_cast<int*>(smem + 32 * SA + kD4Classes * 32 * SB);
  const Geom& g = p.g;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int HC = g.H / g.S, WC = g.W / g.S;
  const int Mc = g.B * HC * WC;
  const int i0 = bx * BI;
  const int JX = g.KW / g.S;
  const int nps = g.N >> 5;                         // 32-deep steps per tap
  const int nsteps = (g.KH / g.S) * JX * nps;
  const FastDiv dhw = p.d_hw[0], dw = p.d_w[0];
  XT_TL(0);
  XT_TL_ROLE(31);
  if (t < BI) {
    const int mc = i0 + t;
    int off = -1;
    if (mc < Mc) {
      const int b = (int)fdiv((uint32_t)mc, dhw), rem = mc - b * (HC * WC);
      const int ty = (int)fdiv((uint32_t)rem, dw), tx = rem - ty * WC;
      off = ((b * g.H + g.S * ty) * g.W + g.S * tx) * g.C;      // class (0,0) pixel of this position
    }
    rowOut[t] = off;
  }
  const int c4 = t & 7, r0 = t >> 3;
  int rowbase[NA], qy[NA], qx[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int mc = i0 + r0 + 32 * i;
    if (mc < Mc) {
      const int b = (int)fdiv((uint32_t)mc, dhw), rem = mc - b * (HC * WC);
      qy[i] = (int)fdiv((uint32_t)rem, dw); qx[i] = rem - qy[i] * WC;
      rowbase[i] = (b * g.OHOW + qy[i] * g.OW + qx[i]) * g.N;
    } else {
      rowbase[i] = 0; qy[i] = -(1 << 28); qx[i] = 0;
    }
  }
  const int cN = r0 * g.N;                          // this thread's weight row (input channel r0 < 32 = C)

  struct Regs { float4 a[NA]; float4 b[kD4Classes]; uint32_t ok; };
  // (buffer loads: 32-bit offsets, taps outside the gradient map read as zeros through the range check -- every
  // instruction of the step that is not an MFMA costs the wave ~5 cycles, see igemm_wgrad_body)
  const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(p.dy, (uint32_t)g.M * (uint32_t)g.N * 4u);
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (uint32_t)g.K * (uint32_t)g.N * 4u);
  auto fetch = [&](int s, Regs& R) {
    const int tap = s / nps, n0 = (s - tap * nps) * 32 + c4 * 4;
    const int jy = tap / JX, jx = tap - jy * JX;
    const int tapoff = (jy * g.OW + jx) * g.N - n0;
    R.ok = 0xfu;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const bool ok = ((unsigned)(qy[i] - jy) < (unsigned)g.OH) && ((unsigned)(qx[i] - jx) < (unsigned)g.OW);
      R.a[i] = buf_load4(rs_dy, ok ? (uint32_t)(rowbase[i] - tapoff) * 4u : kOob, 0);
    }
#pragma unroll
    for (int cls = 0; cls < kD4Classes; ++cls) {
      const int ry = cls / g.S, rx = cls - ry * g.S;
      const int wbase = ((ry + g.S * jy) * g.KW + rx + g.S * jx) * g.C * g.N + n0;
      R.b[cls] = buf_load4(rs_w, (uint32_t)(wbase + cN) * 4u, 0);
    }
  };
  auto stash = [&](const Regs& R) {
    if constexpr (SPLIT) {
      const int sub = c4 >> 1, byte = (c4 & 1) * 8;     // k = 4*c4 + e -> chunk = c4 >> 2, k half = (c4 & 3) >> 1
#pragma unroll
      for (int i = 0; i < NA; ++i)
        split3_store(Ap + (r0 + 32 * i) * 16 + sub * kD4SlotA + byte, 4 * kD4SlotA, R.a[i]);
#pragma unroll
      for (int cls = 0; cls < kD4Classes; ++cls)
        split3_store(Bp + cls * 12 * kD4SlotB + r0 * 16 + sub * kD4SlotB + byte, 4 * kD4SlotB, R.b[cls]);
      return;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = r0 + 32 * i;
      const float4 v = R.a[i];
      As[(c4 * 4 + 0) * SA + r] = v.x;
      As[(c4 * 4 + 1) * SA + r] = v.y;
      As[(c4 * 4 + 2) * SA + r] = v.z;
      As[(c4 * 4 + 3) * SA + r] = v.w;
    }
#pragma unroll
    for (int cls = 0; cls < kD4Classes; ++cls) {
      float* Bc = Bs + cls * 32 * SB;
      Bc[(c4 * 4 + 0) * SB + r0] = R.b[cls].x;
      Bc[(c4 * 4 + 1) * SB + r0] = R.b[cls].y;
      Bc[(c4 * 4 + 2) * SB + r0] = R.b[cls].z;
      Bc[(c4 * 4 + 3) * SB + r0] = R.b[cls].w;
    }
  };
  f32x16 acc[kD4Classes];
#pragma unroll
  for (int cls = 0; cls < kD4Classes; ++cls)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cls][r] = 0.f;
  // one register stage: a step carries 64 MFMAs per wave (4096 cycles), which covers the fetch of the next one --
  // with fp32 MFMAs.  On the bf16 pipes (SPLIT) a step is 1536 cycles = 0.64 us, a fraction of a global round trip
  // under load: the loop then runs at the memory latency (timeline: 3 us per step).  PF4 (nsteps == 4, two workgroups
  // per CU = 256 VGPRs): the operands of ALL four taps are requested up front, the loop only splits, syncs and multiplies.
  Regs R0, R1, R2, R3;
  fetch(0, R0);
  if constexpr (PF4) { fetch(1, R1); fetch(2, R2); fetch(3, R3); }
  XT_TL(1);
  const int kl = lane >> 5, il = lane & 31;
  auto mma = [&]() {
    if constexpr (SPLIT) {
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        bf16x8 a[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          a[pl] = *reinterpret_cast<const bf16x8*>(Ap + (pl * 4 + ch * 2 + kl) * kD4SlotA + (wave * 32 + il) * 16);
#pragma unroll
        for (int cls = 0; cls < kD4Classes; ++cls) {
          bf16x8 b[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            b[pl] = *reinterpret_cast<const bf16x8*>(Bp + (cls * 12 + pl * 4 + ch * 2 + kl) * kD4SlotB + il * 16);
          acc[cls] = mfma_bf16x6(a, b, acc[cls]);
        }
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float a = As[(kk * 2 + kl) * SA + wave * 32 + il];
#pragma unroll
      for (int cls = 0; cls < kD4Classes; ++cls)
        acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[cls * 32 * SB + (kk * 2 + kl) * SB + il], acc[cls], 0, 0, 0);
    }
  };
  if constexpr (PF4) {
    stash(R0); __syncthreads(); XT_TL(2); mma(); __syncthreads();
    stash(R1); __syncthreads(); mma(); __syncthreads();
    stash(R2); __syncthreads(); mma(); __syncthreads();
    stash(R3); __syncthreads(); mma(); __syncthreads();
  } else {
    for (int s = 0; s < nsteps; ++s) {
      stash(R0);
      __syncthreads();
      if (s == 0) XT_TL(2);
      if (s + 1 < nsteps) fetch(s + 1, R0);
      mma();
      __syncthreads();
    }
  }
  XT_TL(3);
  // epilogue: class (ry, rx) of position row i writes pixel rowOut[i] + (ry*W + rx)*C.  A pixel is one contiguous
  // 128-byte row of C = 32 channels, but the accumulator layout has lanes = channels (dword accesses, 2 rows per
  // instruction: 64 loads of the producer activation + 64 stores per lane, 6.5 us of the block's 25).  Each class
  // tile is therefore transposed through LDS (As/Bs are dead after the loop's last barrier; wave-private regions,
  // program order of one wave suffices) so that every lane moves 16 bytes: 4 loads + 4 stores per class.  The
  // producer activations of class c+1 are requested before class c is stored.
  float* tb = smem + wave * (32 * 36);
  const int tr = lane >> 3, tc4 = (lane & 7) * 4;
  int roff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) roff[q] = rowOut[wave * 32 + q * 8 + tr];
  auto class_off = [&](int cls) { const int ry = cls / g.S, rx = cls - ry * g.S; return (ry * g.W + rx) * g.C + tc4; };
  auto load_x = [&](float4 (&xv)[4], int cls) {
    const int coff = class_off(cls);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      xv[q] = *reinterpret_cast<const float4*>(p.x + (roff[q] >= 0 ? (size_t)(roff[q] + coff) : (size_t)0));
  };
  auto store_dx = [&](const float4 (&xv)[4], int cls) {
    const int coff = class_off(cls);
#pragma unroll
    for (int r = 0; r < 16; ++r) tb[((r & 3) + 8 * (r >> 2) + 4 * kl) * 36 + il] = acc[cls][r];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v = *reinterpret_cast<const float4*>(&tb[(q * 8 + tr) * 36 + tc4]);
      v.x *= act_grad(xv[q].x, p.act_prev);
      v.y *= act_grad(xv[q].y, p.act_prev);
      v.z *= act_grad(xv[q].z, p.act_prev);
      v.w *= act_grad(xv[q].w, p.act_prev);
      if (roff[q] >= 0) store4_wt(p.dx, (size_t)(roff[q] + coff), v);
    }
  };
  if (p.xmask) {
    // relu'(x) from the producer's sign mask: 4 bytes per pixel instead of its 128-byte activation row (the x reads
    // were half of this epilogue's HBM traffic: 16 MB next to the 16 MB of dX it writes; probe: 22.7 -> 20.9 us)
    uint32_t mw[4][4];
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const int ry = cls / g.S, rx = cls - ry * g.S;
#pragma unroll
      for (int q = 0; q < 4; ++q) mw[cls][q] = p.xmask[roff[q] >= 0 ? (roff[q] >> 5) + ry * g.W + rx : 0];
    }
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const int coff = class_off(cls);
#pragma unroll
      for (int r = 0; r < 16; ++r) tb[((r & 3) + 8 * (r >> 2) + 4 * kl) * 36 + il] = acc[cls][r];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = *reinterpret_cast<const float4*>(&tb[(q * 8 + tr) * 36 + tc4]);
        const uint32_t bits = mw[cls][q] >> tc4;
        v.x = (bits & 1u) ? v.x : 0.f; v.y = (bits & 2u) ? v.y : 0.f;
        v.z = (bits & 4u) ? v.z : 0.f; v.w = (bits & 8u) ? v.w : 0.f;
        if (roff[q] >= 0) store4_wt(p.dx, (size_t)(roff[q] + coff), v);
      }
    }
    XT_TL(4);
    XT_TL_DRAIN(5);
    return;
  }
  float4 xa[4], xb[4];
  load_x(xa, 0);
  load_x(xb, 1);
  store_dx(xa, 0);
  load_x(xa, 2);
  store_dx(xb, 1);
  load_x(xb, 3);
  store_dx(xa, 2);
  store_dx(xb, 3);
  XT_TL(4);
  XT_TL_DRAIN(5);
}

template <int BI, int BJ, int WI, int WJ, bool X6 = false>
__global__ __launch_bounds__(256) void igemm_dgrad_kernel(const DgradArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[dgrad_smem_floats<BI, BJ, X6>()];
  igemm_dgrad_body<BI, BJ, WI, WJ, X6>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// ------------------------------------------------------------------ weight gradient from staged input ROWS
// (round 3) Weight gradient of a VALID stride-S conv with C = N = 32 and KW = 4 (PpoCnn conv2: 4x4/2, 20x20x32 ->
// 9x9x32) WITHOUT the im2col gather: the LDS-tiled form above streams the KH*KW/S^2-times expanded im2col view of the
// layer input through the L1 (53 MB for a 16.4 MB activation at B = 320, plus dY once per k tile) and that vector-memory
// traffic -- not the matrix pipe -- paced the whole fused backward launch (in-kernel phase probe).
//   workgroup = (sample group g, kernel row ky); wave = kernel column kx  -> one 32 (c) x 32 (n) accumulator tile
//   per wave, kept in registers across the group's samples; reduction = the OH*OW output positions of a sample.
// Per sample the OH input rows ky, ky+S, ... (each W*C contiguous floats: plain 16-byte row copies, no gather) and
// the sample's dY [OH*OW, 32] are staged in LDS once; BOTH MFMA operands are then plain ds_read_b32 of the natural
// layouts: A[c][m] = x[row oy][(S*ox + kx)*32 + c] (32 consecutive floats), B[m][n] = dY[m][n] -- the im2col view is
// a per-lane address (a small per-position offset table in LDS), not data movement.  The two reduction slots of
// v_mfma_f32_32x32x2_f32 take the even / odd positions; positions past the map multiply a zero dY row.
// The next sample's rows are requested into registers before the current one is multiplied.
// Input traffic: every input row is read by the KH/S kernel rows that use it (2x for 4x4/2) instead of KH*KW/S^2 (4x)
// and dY KH times, all as full-line row copies.  Slab layout as igemm_wgrad_body: slab g, rows [ky*KW*32, +KW*32).
constexpr int kWrC = 32, kWrN = 32, kWrKW = 4;
__host__ __device__ constexpr int wrows_smem_floats(int OH, int OW, int W) {
  return OH * (W * kWrC + 4) + 2 * (((OH * OW + 7) >> 3) << 2) * kWrN + 256 + 2 * (((OH * OW + 7) >> 3) << 2);
}
constexpr int kWrMaxSmemFloats = 12 * 1024;     // what the fused kernel instance reserves (>= wrows_smem_floats of the shape)

#ifndef XT_WR_CHAINS
#define XT_WR_CHAINS 4
#endif
constexpr int kWrChains = XT_WR_CHAINS;      // independent accumulator chains per wave (A/B: 1, 2, 4)
constexpr int kWrXQ = 6, kWrDQ = 3;          // per-thread float4 slots of a sample's staging (<= 1536 / 768 float4)
typedef float f4v __attribute__((ext_vector_type(4)));   // (HIP float4 members kept this struct in scratch memory)
struct WrRegs { f4v x[kWrXQ]; f4v d[kWrDQ]; };

__device__ __forceinline__ void wr_fetch(WrRegs& R, const float* __restrict__ xb, const float* __restrict__ db, int t,
                                         int ky, int S, int WC, int rowq, FastDiv d_rowq, int nxq, int ndq) {
#pragma unroll
  for (int i = 0; i < kWrXQ; ++i) {
    const int q = t + 256 * i;
    const int oy = (int)fdiv((uint32_t)q, d_rowq), c4 = q - oy * rowq;
    R.x[i] = *reinterpret_cast<const f4v*>(xb + (q < nxq ? (oy * S + ky) * WC + c4 * 4 : 0));
  }
#pragma unroll
  for (int i = 0; i < kWrDQ; ++i) {
    const int q = t + 256 * i;
    R.d[i] = *reinterpret_cast<const f4v*>(db + (q < ndq ? q * 4 : 0));
  }
}
__device__ __forceinline__ void wr_stash(const WrRegs& R, float* xs, float* ds, int t, int RS, int rowq, FastDiv d_rowq,
                                         int nxq, int ndq) {
#pragma unroll
  for (int i = 0; i < kWrXQ; ++i) {
    const int q = t + 256 * i;
    const int oy = (int)fdiv((uint32_t)q, d_rowq), c4 = q - oy * rowq;
    if (q < nxq) *reinterpret_cast<f4v*>(xs + oy * RS + c4 * 4) = R.x[i];
  }
#pragma unroll
  for (int i = 0; i < kWrDQ; ++i) {
    const int q = t + 256 * i;
    if (q < ndq) *reinterpret_cast<f4v*>(ds + q * 4) = R.d[i];
  }
}

__device__ __forceinline__ void wgrad_rows_body(const WgradArgs& p, const int group, const int ky, const int per_group,
                                                float* smem) {
  const Geom& g = p.g;
  const int t = threadIdx.x, lane = t & 63, kx = t >> 6;
  const int il = lane & 31, kl = lane >> 5;
  const int WC = g.W * kWrC;
  const int RS = WC + 4;                               // LDS row stride (floats), 16-byte aligned rows
  const int half = ((g.OHOW + 7) >> 3) << 2;           // positions per reduction slot, a multiple of 4 (zero rows pad)
  float* xs = smem;                                    // [OH][RS]
  float* ds = smem + g.OH * RS;                        // [2*half][32]; rows >= OHOW are zero
  float* red = ds + 2 * half * kWrN;                   // [8][32] bias partials
  int* tab = reinterpret_cast<int*>(red + 256);        // [2][half] A-operand byte offsets per position
  const float* x = static_cast<const float*>(p.in);
  const int s_beg = group * per_group, s_end = min(g.B, s_beg + per_group);
  XT_TL(0);
  XT_TL_ROLE(21);
  if (s_beg >= s_end) return;                          // (block-uniform)

  const int rowq = WC / 4;                             // float4 per input row
  const int nxq = g.OH * rowq, ndq = g.OHOW * (kWrN / 4);
  const FastDiv d_rowq = p.d_rowq;
  // zero rows of the dY tile (positions OHOW .. 2*half-1), written once
  for (int e = t; e < (2 * half - g.OHOW) * kWrN; e += 256) ds[g.OHOW * kWrN + e] = 0.f;

  // FOUR independent accumulator chains (position pair u of every step): a dependent v_mfma_f32_32x32x2_f32 issues
  // only ~160 cycles after its predecessor from a lone wave (timeline: 164 cycles per MFMA with one chain, 64 is the
  // pipe rate); summed in fixed order at the end
  f32x16 acc4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc4[u][r] = 0.f;
  const bool do_bias = (ky == 0);
  const int bn = t & 31, bg = t >> 5;
  float bsum = 0.f;

  // A-operand byte offset of every position (same for every sample): [slot kl][half] ints; the two reduction slots
  // take the even / odd positions (lane groups {0-31} / {32-63} of a ds_read_b32 never conflict with each other);
  // positions past the map read pixel 0 against a zero dY row
  for (int e = t; e < 2 * half; e += 256) {
    const int slot = e >= half ? 1 : 0, pos = 2 * (e - slot * half) + slot;
    const int oy = (int)fdiv((uint32_t)pos, g.d_ow), ox = pos - oy * g.OW;
    tab[e] = pos < g.OHOW ? (oy * RS + ox * g.S * kWrC) * 4 : 0;
  }

  // two register sets: sample s+2 is requested while sample s is multiplied (one sample's MFMAs, ~1.2 us, are
  // shorter than a global round trip under load: with a single set the loop ran at the memory latency, 3 us per sample)
  WrRegs RA, RB;
  auto fetch_s = [&](WrRegs& R, int sidx) __attribute__((always_inline)) {
    const int sn = sidx < s_end ? sidx : s_end - 1;          // (unconditional, clamped)
    wr_fetch(R, x + (size_t)sn * g.HWC, p.dy + (size_t)sn * g.OHOW * kWrN, t, ky, g.S, WC, rowq, d_rowq, nxq, ndq);
  };
  fetch_s(RA, s_beg);
  fetch_s(RB, s_beg + 1);
  XT_TL(1);
  const char* ap = reinterpret_cast<const char*>(xs + kx * kWrC + il);
  const float* bp = ds + kl * kWrN + il;                 // position 2*j + kl -> row stride 2*32 floats per j
  const int4* tp = reinterpret_cast<const int4*>(tab + kl * half);
  const int nst = half >> 2;
  for (int s = s_beg; s < s_end; ++s) {
    const bool even = ((s - s_beg) & 1) == 0;            // block-uniform
    __syncthreads();                       // everybody is done reading the previous sample's tiles
    if (even) wr_stash(RA, xs, ds, t, RS, rowq, d_rowq, nxq, ndq); else wr_stash(RB, xs, ds, t, RS, rowq, d_rowq, nxq, ndq);
    __syncthreads();
    if (s == s_beg) XT_TL(2);
    if (s + 2 < s_end) { if (even) fetch_s(RA, s + 2); else fetch_s(RB, s + 2); }
#ifdef XT_TL_EXPERIMENT
    if (s == s_beg) XT_TL(3);
#endif
    // four position pairs per step, two register sets (no copies): the eight operand reads of step i+1 are ISSUED
    // before the four MFMAs of step i and the offset quad of step i+2 before that (clamped, unconditional).  The
    // sched_barriers pin that order: left alone, the scheduler sinks every read next to its MFMA and the wave -- the
    // only one this workgroup has on its SIMD -- exposes a full LDS round trip per MFMA (measured: 26.5 us launch).
    auto load4 = [&](float (&av)[4], float (&bv)[4], const int4 o, int st) __attribute__((always_inline)) {
      av[0] = *reinterpret_cast<const float*>(ap + o.x); av[1] = *reinterpret_cast<const float*>(ap + o.y);
      av[2] = *reinterpret_cast<const float*>(ap + o.z); av[3] = *reinterpret_cast<const float*>(ap + o.w);
#pragma unroll
      for (int u = 0; u < 4; ++u) bv[u] = bp[(st * 4 + u) * 2 * kWrN];
    };
    auto mma4 = [&](const float (&av)[4], const float (&bv)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        acc4[u % kWrChains] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc4[u % kWrChains], 0, 0, 0);
    };
    const int last = nst - 1;
    float a0[4], b0[4], a1[4], b1[4];
    int4 oa = tp[last < 1 ? last : 1], ob = tp[last < 2 ? last : 2];
    load4(a0, b0, tp[0], 0);
    // Issue order inside a half step (sched_group_barrier): MFMA, then two VALU (next A addresses) and two LDS reads,
    // four times.  A lone wave issues in order and stalls at an MFMA until the pipe is free, so only what sits BETWEEN
    // two MFMAs runs in the shadow of the first; with the 17 address / LDS instructions of a half step behind its four
    // MFMAs the loop measured 115 cycles per MFMA (64 is the pipe rate).
#define XT_WR_INTERLEAVE()                                  \
  do {                                                      \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {      \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    \
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);    \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    \
    }                                                       \
  } while (0)
    for (int i = 0; i + 1 < nst; i += 2) {      // steps i (set 0) and i + 1 (set 1); an odd last step follows the loop
      const int i2 = i + 2 < last ? i + 2 : last, i3 = i + 3 < last ? i + 3 : last, i4 = i + 4 < last ? i + 4 : last;
      load4(a1, b1, oa, i + 1);
      oa = tp[i3];
      mma4(a0, b0);
      XT_WR_INTERLEAVE();
      __builtin_amdgcn_sched_barrier(0);
      load4(a0, b0, ob, i2);
      ob = tp[i4];
      mma4(a1, b1);
      XT_WR_INTERLEAVE();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (nst & 1) mma4(a0, b0);
#undef XT_WR_INTERLEAVE
#ifdef XT_TL_EXPERIMENT
    if (s == s_beg) XT_TL(4);
#endif
    if (do_bias) {
      for (int m = bg; m < g.OHOW; m += 8) bsum += ds[m * kWrN + bn];
    }
  }
#ifndef XT_TL_EXPERIMENT
  XT_TL(3);
#endif
  float* out = p.out + (size_t)group * ((size_t)(g.K + 1) * kWrN);
  const int k0 = (ky * kWrKW + kx) * kWrC;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = (r & 3) + 8 * (r >> 2) + 4 * kl;
    out[(size_t)(k0 + c) * kWrN + il] = (acc4[0][r] + acc4[1][r]) + (acc4[2][r] + acc4[3][r]);
  }
  if (do_bias) {
    __syncthreads();
    red[bg * kWrN + bn] = bsum;
    __syncthreads();
    if (t < kWrN) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += red[q * kWrN + t];
      out[(size_t)g.K * kWrN + t] = sum;
    }
  }
#ifndef XT_TL_EXPERIMENT
  XT_TL(4);
#endif
  XT_TL_DRAIN(5);
}

// ------------------------------------------------------------------ fused backward of one layer
// One launch = weight-gradient blocks + input-gradient blocks (+ the head weight-gradient blocks for the
// last trunk layer).  All three only consume d(pre-activation) of this layer, so running them side by side
// removes two dependent-launch start-ups per layer and fills the CUs that a single short kernel leaves idle.
struct BwdLayerArgs {
  WgradArgs wg;
  DgradArgs dg;
  DDgradArgs ddg;            // register-direct input gradient (xt_direct_dev.h), used when dg_direct != 0
  int dg_direct;
  int dg_xcd;                // LDS-tiled input-gradient blocks in XCD-contiguous order
  HeadWgArgs hw;
  int wg_gx, wg_gy, wg_gz;   // wgrad grid
  int dg_gx, dg_gy, dg_gz;   // dgrad grid
  int n_wg, n_dg, n_hw;      // block counts (n_hw may be 0)
};

// HALO = true: the input-gradient blocks are halo_dgrad_body, selected at COMPILE time.  The register allocation of
// a kernel is the maximum over all of its paths: the generic form needs 144 VGPR (LDS-tiled dgrad) + 32 AGPR
// (register-direct dgrad) = two workgroups per CU, this one three.
template <int WBI, int WBJ, int WWI, int WWJ, bool WPAD, int DBI, int DBJ, int DWI, int DWJ, int D4 = 0, int HALO = 0,
          bool DX6 = false, int WROWS = 0, int WX6 = 0>
__global__ __launch_bounds__(256, WROWS ? 2 : 3) void igemm_bwd_layer_kernel(const BwdLayerArgs p) {
  constexpr int SMD = HALO == 5 ? 18 * 1024 : HALO == 3 ? 8 * 1024 : HALO == 2 ? 11 * 1024 : HALO == 1 ? 9 * 1024 : dgrad_smem_floats<DBI, DBJ, DX6>();
  constexpr int SMW = (WROWS == 1 || WROWS == 2) ? kWrMaxSmemFloats : wgrad_smem_floats<WBI, WBJ, WPAD, WX6>();
  constexpr int SM0 = SMW > SMD ? SMW : SMD;
  constexpr int SM = (D4 && dgrad4_smem_floats<D4 == 2>() > SM0) ? dgrad4_smem_floats<D4 == 2>() : SM0;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  int b = blockIdx.x;
#ifdef XT_TL_EXPERIMENT
  if (b < p.n_dg && p.dg_direct == 99) return;      // experiment: weight-gradient blocks alone
#endif
  if (b < p.n_dg) {                       // dgrad first: it is on the critical path of the next layer
    if constexpr (D4 != 0) {              // stride-2 conv: the four parity classes of a position tile in one block
      igemm_dgrad4_body<D4 == 2, (WROWS >= 2)>(p.dg, b, smem);
      return;
    }
    if constexpr (HALO == 5) {            // ... input AND weight gradient per sample (the launch has no weight-gradient blocks)
      s2c16_bwd_body(p.ddg, p.wg.out, p.n_dg, (uint32_t)b, smem);
      return;
    }
    if constexpr (HALO == 3) {            // stride-2 4x4, 16 input channels: one sample per workgroup, one class per wave
      s2c16_dgrad_body<1>(p.ddg, (uint32_t)b, smem);
      return;
    }
    if constexpr (HALO == 1 || HALO == 2) {
      halo_dgrad_body<HALO == 2>(p.ddg, (uint32_t)b, (uint32_t)p.n_dg, smem);
      return;
    }
    if (p.dg_direct == 4) {               // stride-1, small map: 64-row tiles with the dY halo staged in LDS
      halo_dgrad_body<false>(p.ddg, (uint32_t)b, (uint32_t)p.n_dg, smem);
      return;
    }
    if (p.dg_direct == 3) {               // the same with 64-row tiles (weight operand shared by two row tiles)
      direct_dgrad_body<2, 1, 4>(p.ddg, (uint32_t)b, (uint32_t)p.n_dg, smem);
      return;
    }
    if (p.dg_direct) {                    // single-column tiles with a long reduction: 4 independent waves per tile
      direct_dgrad_body<1, 1, 4>(p.ddg, (uint32_t)b, (uint32_t)p.n_dg, smem);
      return;
    }
    // (the M tiles of one channel tile stream the same W^T slice: XCD-contiguous order, as the forward)
    if (p.dg_xcd) b = (int)xcd_chunk((uint32_t)b, (uint32_t)p.n_dg);
    const int bx = b % p.dg_gx, r = b / p.dg_gx;
    igemm_dgrad_body<DBI, DBJ, DWI, DWJ, DX6, (WROWS == 3 ? 8 : 0)>(p.dg, bx, r % p.dg_gy, r / p.dg_gy, smem);
    return;
  }
  b -= p.n_dg;
  if (b < p.n_wg) {
    // XCD-contiguous order: the k tiles of one m range gather overlapping im2col lines of the same activation
    // rows; consecutive block ids land on different XCDs (round-robin dispatch, private L2s), which made every
    // XCD fetch the layer input once per k tile (PMC: FETCH 2x35 MB for a 16 MB input)
    b = (int)xcd_chunk((uint32_t)b, (uint32_t)p.n_wg);
    if constexpr (WROWS == 1 || WROWS == 2) {   // staged-rows weight gradient: (sample group, kernel row) per workgroup
      wgrad_rows_body(p.wg, b / p.wg_gx, b % p.wg_gx, p.wg.mchunk, smem);
      return;
    }
    const int bx = b % p.wg_gx, r = b / p.wg_gx;
    igemm_wgrad_body<WBI, WBJ, WWI, WWJ, false, WPAD, (WROWS == 3), WX6>(p.wg, bx, r % p.wg_gy, r / p.wg_gy, smem);
    return;
  }
  b -= p.n_wg;
  heads_wgrad_partial_body(p.hw, b % p.hw.gx, b / p.hw.gx, smem);
}

XT_TL_SETTER(igemm)

// ------------------------------------------------------------------ host launchers
static inline int pick_ksplit_chunk(int K, int split, int* chunk) {
  int steps = (K + 31) / 32;
  int per = (steps + split - 1) / split;
  *chunk = per * 32;
  return (steps + per - 1) / per;   // effective split count
}

int launch_conv1_fwd_bf16x3(const xt_conv_geom*, const xt_input_xform*, int, const void*, const int32_t*, const float*,
                            const float*, float*, hipStream_t, uint32_t*, int*);
int launch_conv1_wgrad_bf16x3(const xt_conv_geom*, const xt_input_xform*, int, const void*, const int32_t*,
                              const float*, float*, float*, int, int*, hipStream_t);
int launch_conv1_same_fwd(const xt_conv_geom*, const xt_input_xform*, int, const void*, const int32_t*, const float*,
                          const float*, float*, hipStream_t);
int launch_conv1_same_wgrad(const xt_conv_geom*, const xt_input_xform*, int, const void*, const int32_t*, const float*,
                            float*, float*, int, int*, hipStream_t);
bool plan_dgrad_direct_fused(const Geom&, DDgradArgs*, int*);
int launch_dgrad_direct(const xt_conv_geom*, int, const float*, const float*, const float*, int, float*, hipStream_t);
int launch_fwd_direct(const xt_conv_geom*, const xt_input_xform*, int, const void*, const int32_t*, const float*,
                      const float*, float*, float*, int, hipStream_t, int*);

static bool use_kg2() { return tuning().fwd_two_groups != 0; }
static bool use_bf16x3() { return tuning().conv1_bf16x3 != 0; }

static int fill_class_divs(const Geom& g, DgradArgs* a) {
  XT_REQUIRE(g.S * g.S <= kMaxClasses, "igemm dgrad: stride %d not supported (max 4)", g.S);
  for (int cls = 0; cls < g.S * g.S; ++cls) {
    const int ry = cls / g.S, rx = cls % g.S;
    const int cy0 = ((ry - g.PT) % g.S + g.S) % g.S, cx0 = ((rx - g.PL) % g.S + g.S) % g.S;
    const int HC = cy0 < g.H ? (g.H - cy0 + g.S - 1) / g.S : 0;
    const int WC = cx0 < g.W ? (g.W - cx0 + g.S - 1) / g.S : 0;
    XT_REQUIRE((long long)g.B * HC * WC * HC * WC < (1ll << 32), "igemm dgrad: class extent too large for the fast divide");
    a->d_hw[cls] = make_fastdiv((uint32_t)(HC * WC > 0 ? HC * WC : 1));
    a->d_w[cls] = make_fastdiv((uint32_t)(WC > 0 ? WC : 1));
  }
  return 0;
}

int launch_fwd(const xt_conv_geom* cg, const xt_input_xform* xf, int B, const void* in, const int32_t* idx,
               const float* w, const float* bias, float* y, float* partial, int ksplit, hipStream_t st,
               int* deferred_ksplit, uint32_t* relu_mask, int* mask_written) {
  if (mask_written) *mask_written = 0;
  if (use_bf16x3()) {     // uint8 first layer: exact 3-way bf16 split on the bf16 matrix cores
    const int rc = launch_conv1_fwd_bf16x3(cg, xf, B, in, idx, w, bias, y, st, relu_mask, mask_written);
    if (rc >= 0) { last_arith() = XT_ARITH_BF16X3; if (deferred_ksplit) *deferred_ksplit = 1; return rc; }
    const int rs = launch_conv1_same_fwd(cg, xf, B, in, idx, w, bias, y, st);      // ImpalaCnnOpt's first layers
    if (rs >= 0) { last_arith() = XT_ARITH_BF16X3; if (deferred_ksplit) *deferred_ksplit = 1; return rs; }
  }
  FwdArgs a;
  if (int rc = make_geom(cg, xf, B, &a.g)) return rc;
  // register-direct kernel (xt_direct.hip) when the shape is inside its envelope -- except un-padded N <= 32 layers when
  // the LDS-tiled bf16x6 forward is on (PpoCnn conv2: 7.53 / 7.59 -> 7.40 / 7.49 ms per update; with SAME padding the
  // direct kernel stays ahead: ImpalaCnnOpt conv2 7.1 vs 11.3 us at 128 frames, 28.6 vs 31.1 at 1000)
  const bool tiled_first = tuning().fwd_tiled_valid && tuning().bf16x6 && !(xf && xf->is_u8) && a.g.N <= 32 &&
                           a.g.K >= 256 && !is_padded(a.g);
  if (!tiled_first) {
    int ks = 1;
    const int rc = launch_fwd_direct(cg, xf, B, in, idx, w, bias, y, partial, ksplit, st, &ks);
    if (rc > 0) return rc;
    if (rc == 0) {
      last_arith() = XT_ARITH_FP32;
      if (deferred_ksplit) *deferred_ksplit = ks;
      if (ks > 1 && !deferred_ksplit) {
        const int MN = a.g.M * a.g.N;
        hipLaunchKernelGGL(splitk_finish_kernel, dim3((MN / 4 + 255) / 256), dim3(256), 0, st,
                           partial, bias, y, MN, a.g.N, ks, a.g.act);
        XT_LAUNCH_CHECK();
      }
      return 0;
    }
  }
  const bool u8 = xf && xf->is_u8;
  a.in = in; a.idx = idx; a.w = w; a.bias = bias;
  if (ksplit < 1) ksplit = 1;
  int chunk;
  ksplit = pick_ksplit_chunk(a.g.K, ksplit, &chunk);
  XT_REQUIRE(ksplit == 1 || partial != nullptr, "xt_layer_fwd: ksplit>1 needs a partial buffer");
  a.ksplit = ksplit; a.kchunk = chunk;
  a.y = ksplit == 1 ? y : partial;
  const int M = a.g.M, N = a.g.N;
  a.xcd_chunked = (tuning().fwd_xcd_chunk != 0 && (N > (N <= 32 ? 32 : 64) || ksplit > 1)) ? 1 : 0;
  const bool pad = is_padded(a.g);
  // two wave groups per block when the launch cannot fill the chip and the step chain is long
  const int nblk = (N <= 32 ? ((M + 127) / 128) * ((N + 31) / 32) : ((M + 63) / 64) * ((N + 63) / 64)) * ksplit;
  const bool kg2 = use_kg2() && nblk <= 320 && chunk >= 8 * 32;
#define XT_FWD2(BI, BJ, WI, WJ, KGV)                                                                        \
  do {                                                                                                      \
    dim3 grid((M + BI - 1) / BI, (N + BJ - 1) / BJ, ksplit);                                                \
    dim3 blk(256 * KGV);                                                                                    \
    if (u8 && pad) hipLaunchKernelGGL((igemm_fwd_kernel<BI, BJ, WI, WJ, true, true, KGV>), grid, blk, 0, st, a);     \
    else if (u8) hipLaunchKernelGGL((igemm_fwd_kernel<BI, BJ, WI, WJ, true, false, KGV>), grid, blk, 0, st, a);      \
    else if (pad) hipLaunchKernelGGL((igemm_fwd_kernel<BI, BJ, WI, WJ, false, true, KGV>), grid, blk, 0, st, a);     \
    else hipLaunchKernelGGL((igemm_fwd_kernel<BI, BJ, WI, WJ, false, false, KGV>), grid, blk, 0, st, a);             \
  } while (0)
#define XT_FWD(BI, BJ, WI, WJ) do { if (kg2) XT_FWD2(BI, BJ, WI, WJ, 2); else XT_FWD2(BI, BJ, WI, WJ, 1); } while (0)
#define XT_FWD6(BI, BJ, WI, WJ, KGV)                                                                        \
  do {                                                                                                      \
    dim3 grid((M + BI - 1) / BI, (N + BJ - 1) / BJ, ksplit);                                                \
    if (pad) hipLaunchKernelGGL((igemm_fwd_kernel<BI, BJ, WI, WJ, false, true, KGV, true>), grid, dim3(256 * KGV), 0, st, a);  \
    else hipLaunchKernelGGL((igemm_fwd_kernel<BI, BJ, WI, WJ, false, false, KGV, true>), grid, dim3(256 * KGV), 0, st, a);     \
  } while (0)
  const bool x6 = !u8 && tuning().bf16x6 != 0;
  last_arith() = x6 ? XT_ARITH_BF16X6 : XT_ARITH_FP32;
  // steps per wave group of the two-group form: all of them in flight when there are at most 8 (fwd_prefetch_all)
  const int nst2 = ((chunk + 31) / 32 + 1) / 2;
#define XT_FWD6N(BI, BJ, WI, WJ, NSTV)                                                                      \
  hipLaunchKernelGGL((igemm_fwd_kernel<BI, BJ, WI, WJ, false, false, 2, true, NSTV>),                       \
                     dim3((M + BI - 1) / BI, (N + BJ - 1) / BJ, ksplit), dim3(512), 0, st, a)
  const bool all = x6 && kg2 && !pad && tuning().fwd_prefetch_all != 0;
  if (all && N > 32 && nst2 <= 4) XT_FWD6N(64, 64, 2, 2, 4);
  else if (all && N > 32 && nst2 == 5) XT_FWD6N(64, 64, 2, 2, 5);
  else if (all && N > 32 && nst2 <= 8) XT_FWD6N(64, 64, 2, 2, 8);
  else if (all && N <= 32 && nst2 <= 4) XT_FWD6N(128, 32, 4, 1, 4);
  else if (all && N <= 32 && nst2 <= 8) XT_FWD6N(128, 32, 4, 1, 8);
  else
#undef XT_FWD6N
  if (x6 && kg2 && tuning().fwd_four_groups && nblk <= 256 && chunk >= 16 * 32) {      // <= one block per CU, >= 4 steps per group (shorter chains: no gain)
    if (N > 32) XT_FWD6(64, 64, 2, 2, 4); else XT_FWD6(128, 32, 4, 1, 4);
  } else
  if (x6 && N > 32) { if (kg2) XT_FWD6(64, 64, 2, 2, 2); else XT_FWD6(64, 64, 2, 2, 1); }
  else if (x6) { if (kg2) XT_FWD6(128, 32, 4, 1, 2); else XT_FWD6(128, 32, 4, 1, 1); }
  else if (N <= 32) XT_FWD(128, 32, 4, 1); else XT_FWD(64, 64, 2, 2);
#undef XT_FWD6
#undef XT_FWD
#undef XT_FWD2
  XT_LAUNCH_CHECK();
  if (deferred_ksplit) *deferred_ksplit = ksplit;     // caller sums the partials itself (fused head kernel)
  if (ksplit > 1 && !deferred_ksplit) {
    const int MN = M * N;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((MN / 4 + 255) / 256), dim3(256), 0, st,
                       partial, bias, y, MN, N, ksplit, a.g.act);
    XT_LAUNCH_CHECK();
  }
  return 0;
}

int launch_wgrad(const xt_conv_geom* cg, const xt_input_xform* xf, int B, const void* in, const int32_t* idx,
                 const float* dy, float* dwb, float* slabs, int msplit, hipStream_t st, int reduce_now,
                 int* msplit_out, int slab_cap) {
  if (use_bf16x3() && slabs && !reduce_now && slab_cap >= B) {   // uint8 first layer: one slab per frame stack
    int ms = 0;
    const int rc = launch_conv1_wgrad_bf16x3(cg, xf, B, in, idx, dy, dwb, slabs, slab_cap, &ms, st);
    if (rc >= 0) { last_arith() = XT_ARITH_BF16X3; if (msplit_out) *msplit_out = ms; return rc; }
  }
  if (use_bf16x3() && slabs && !reduce_now) {
    int ms = 0;
    const int rc = launch_conv1_same_wgrad(cg, xf, B, in, idx, dy, dwb, slabs, slab_cap, &ms, st);
    if (rc >= 0) { last_arith() = XT_ARITH_BF16X3; if (msplit_out) *msplit_out = ms; return rc; }
  }
  WgradArgs a;
  if (int rc = make_geom(cg, xf, B, &a.g)) return rc;
  last_arith() = XT_ARITH_FP32;
  const bool u8 = xf && xf->is_u8;
  a.in = in; a.idx = idx; a.dy = dy;
  if (msplit < 1) msplit = 1;
  int chunk;
  msplit = pick_ksplit_chunk(a.g.M, msplit, &chunk);
  XT_REQUIRE(msplit == 1 || slabs != nullptr, "xt_layer_wgrad: msplit>1 needs a slab buffer");
  a.msplit = msplit; a.mchunk = chunk;
  a.out = msplit == 1 ? dwb : slabs;
  a.sq_out = nullptr; a.sq_gx = 0;
  const int K = a.g.K, N = a.g.N;
  const bool pad = is_padded(a.g);
#define XT_WG(BI, BJ, WI, WJ)                                                                               \
  do {                                                                                                      \
    dim3 grid((K + BI - 1) / BI, (N + BJ - 1) / BJ, msplit);                                                \
    if (u8 && pad) hipLaunchKernelGGL((igemm_wgrad_kernel<BI, BJ, WI, WJ, true, true>), grid, dim3(256), 0, st, a);  \
    else if (u8) hipLaunchKernelGGL((igemm_wgrad_kernel<BI, BJ, WI, WJ, true, false>), grid, dim3(256), 0, st, a);   \
    else if (pad) hipLaunchKernelGGL((igemm_wgrad_kernel<BI, BJ, WI, WJ, false, true>), grid, dim3(256), 0, st, a);  \
    else hipLaunchKernelGGL((igemm_wgrad_kernel<BI, BJ, WI, WJ, false, false>), grid, dim3(256), 0, st, a);          \
  } while (0)
  if (N <= 32) XT_WG(128, 32, 4, 1); else XT_WG(64, 64, 2, 2);
#undef XT_WG
  XT_LAUNCH_CHECK();
  if (msplit_out) *msplit_out = msplit;
  if (msplit > 1 && reduce_now) {
    const int count = (K + 1) * N;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((count / 4 + 255) / 256), dim3(256), 0, st, slabs, dwb, count, msplit);
    XT_LAUNCH_CHECK();
  }
  return 0;
}

int launch_dgrad(const xt_conv_geom* cg, int B, const float* dy, const float* w, const float* x, int act_prev,
                 float* dx, hipStream_t st) {
  {
    const int rc = launch_dgrad_direct(cg, B, dy, w, x, act_prev, dx, st);
    if (rc >= 0) return rc;
  }
  DgradArgs a;
  if (int rc = make_geom(cg, nullptr, B, &a.g)) return rc;
  a.dy = dy; a.w = w; a.x = x; a.dx = dx; a.act_prev = act_prev; a.xmask = nullptr;
  const Geom& g = a.g;
  XT_REQUIRE((long long)g.M * g.N * 4 < (1ll << 31), "dgrad: gradient tensors of 2 GiB or more are not supported (batch %d)", B);
  if (int rc = fill_class_divs(g, &a)) return rc;
  const int hc = (g.H + g.S - 1) / g.S, wc = (g.W + g.S - 1) / g.S;   // upper bound on class extent
  const int mc = B * hc * wc;
  const bool x6 = tuning().bf16x6 != 0;
  last_arith() = x6 ? XT_ARITH_BF16X6 : XT_ARITH_FP32;
  if (g.C <= 32) {
    dim3 grid((mc + 127) / 128, (g.C + 31) / 32, g.S * g.S);
    if (x6) hipLaunchKernelGGL((igemm_dgrad_kernel<128, 32, 4, 1, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_dgrad_kernel<128, 32, 4, 1>), grid, dim3(256), 0, st, a);
  } else {
    dim3 grid((mc + 63) / 64, (g.C + 63) / 64, g.S * g.S);
    if (x6) hipLaunchKernelGGL((igemm_dgrad_kernel<64, 64, 2, 2, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_dgrad_kernel<64, 64, 2, 2>), grid, dim3(256), 0, st, a);
  }
  XT_LAUNCH_CHECK();
  return 0;
}

// wgrad (fp32 input) + dgrad (+ head wgrad) of one non-first layer in ONE launch.
// x_grad (may be null = x_in): what the input gradient's activation-derivative epilogue reads -- the producer's
// PRE-activation when its activation is not monotonic (act_needs_preact), else its output x_in
int launch_bwd_layer(const xt_conv_geom* cg, int B, const float* x_in, const float* dy, const float* w,
                     int act_prev, float* dx, float* dwb, float* slabs, int msplit, const HeadWgArgs* hw,
                     int* msplit_out, hipStream_t st, const uint32_t* xmask, int slab_cap, const float* x_grad,
                     float* sq_partials, int* npre_out) {
  if (!x_grad) x_grad = x_in;
  if (npre_out) *npre_out = 0;
  BwdLayerArgs a;
  if (int rc = make_geom(cg, nullptr, B, &a.wg.g)) return rc;
  a.dg.g = a.wg.g;
  const Geom& g = a.wg.g;
  // (the buffer loads of the backward kernels address their tensors with 32-bit byte offsets, 2^31 = "out of range")
  XT_REQUIRE((long long)g.M * g.N * 4 < (1ll << 31) && (long long)g.B * g.HWC * 4 < (1ll << 31),
             "bwd_layer: activation / gradient tensors of 2 GiB or more are not supported (batch %d)", B);
  // ---- wgrad part
  a.wg.in = x_in; a.wg.idx = nullptr; a.wg.dy = dy;
  a.wg.sq_out = nullptr; a.wg.sq_gx = 0;
  if (msplit < 1) msplit = 1;
  int chunk;
  msplit = pick_ksplit_chunk(g.M, msplit, &chunk);
  XT_REQUIRE(msplit == 1 || slabs != nullptr, "bwd_layer: msplit>1 needs a slab buffer");
  a.wg.msplit = msplit; a.wg.mchunk = chunk;
  a.wg.out = msplit == 1 ? dwb : slabs;
  if (msplit_out) *msplit_out = msplit;
  const bool wsmall = g.N <= 32;
  a.wg_gx = wsmall ? (g.K + 127) / 128 : (g.K + 63) / 64;
  a.wg_gy = wsmall ? (g.N + 31) / 32 : (g.N + 63) / 64;
  a.wg_gz = msplit;
  a.n_wg = a.wg_gx * a.wg_gy * a.wg_gz;
  // ---- dgrad part
  a.dg.dy = dy; a.dg.w = w; a.dg.x = x_grad; a.dg.dx = dx; a.dg.act_prev = act_prev; a.dg.xmask = nullptr;
  if (int rc = fill_class_divs(g, &a.dg)) return rc;
  const int hc = (g.H + g.S - 1) / g.S, wc = (g.W + g.S - 1) / g.S;
  const int mc = B * hc * wc;
  const bool dsmall = g.C <= 32;
  a.dg_gx = dsmall ? (mc + 127) / 128 : (mc + 63) / 64;
  a.dg_gy = dsmall ? (g.C + 31) / 32 : (g.C + 63) / 64;
  a.dg_gz = g.S * g.S;
  a.n_dg = a.dg_gx * a.dg_gy * a.dg_gz;
  a.dg_direct = 0;
  a.dg_xcd = tuning().fwd_xcd_chunk != 0 ? 1 : 0;
  {
    const int no_d4 = tuning().dgrad_all_classes ? 0 : 1;
    if (!no_d4 && g.S == 2 && g.KH % 2 == 0 && g.KW % 2 == 0 && g.H % 2 == 0 && g.W % 2 == 0 && g.PT == 0 &&
        g.PL == 0 && g.C == 32 && g.N == 32 && (g.OH - 1) * g.S + g.KH <= g.H && (g.OW - 1) * g.S + g.KW <= g.W) {
      a.dg_direct = 2;
      a.n_dg = (B * (g.H / 2) * (g.W / 2) + 127) / 128;
      if (act_prev == XT_ACT_RELU) a.dg.xmask = xmask;       // (C == 32: one mask word per pixel)
    }
  }
  if (a.dg_direct == 0) {
    int nblk = 0;
    a.ddg.deep = 0;
    if (plan_dgrad_direct_fused(g, &a.ddg, &nblk)) {
      a.ddg.deep = tuning().bwd_deep_prefetch != 0 ? 1 : 0;
      a.ddg.dy = dy; a.ddg.w = w; a.ddg.x = x_grad; a.ddg.dx = dx; a.ddg.act_prev = act_prev;
      a.dg_direct = 1;
      a.n_dg = nblk;
      const int ti2 = tuning().dgrad_tile64;   // 0: 32-row tiles (A/B; measured 30.4 vs 27.7 us for conv3 at B=320)
      if (ti2 && g.S == 1 && nblk > 512) { // 64-row tiles: half the blocks, the weight operand shared by two row tiles
        const int mc = B * g.H * g.W;
        a.ddg.mt = (mc + 63) / 64;
        a.n_dg = a.ddg.mt * a.ddg.ct;
        a.dg_direct = 3;
        const int halo = tuning().dgrad_halo;   // 0: register-direct dY gather instead of the LDS halo (A/B)
        const int nsamp = 63 / (g.H * g.W) + 2;
        if (halo && (size_t)(nsamp * g.OHOW + 1) * (g.N + 4) * 4 <= 36 * 1024 && g.N + 4 <= 256) a.dg_direct = 4;
      }
    }
  }
  // ---- head wgrad part
  a.n_hw = 0;
  if (hw) { a.hw = *hw; a.n_hw = hw->gx * hw->nchunk; }
  else { a.hw.gx = 1; a.hw.nchunk = 0; }
  // The halo input gradient has its own kernel instance (tuning.bwd_own_instance = 0: the generic one): three workgroups per CU
  // instead of two.  With it, a launch that is only a little larger than the 768 co-resident workgroups is cut to
  // one round (tuning.bwd_fit_slots, 0 = off): the surplus weight-gradient blocks otherwise start when the first
  // input-gradient blocks END and the launch takes two block lifetimes.  Measured for conv3 at B=320: generic 26.3,
  // own instance 24.4, own instance + one round 22.6 us (with the register-direct dY gather both made it SLOWER:
  // a third co-resident workgroup thrashed the L1 that gather depends on).
  const int spec = tuning().bwd_own_instance, fit = tuning().bwd_fit_slots;
  const bool halo_inst = a.dg_direct == 4 && spec && !wsmall && !is_padded(g);
  if (halo_inst) {
    const int tiles = a.wg_gx * a.wg_gy;
    const int room = fit - a.n_dg - a.n_hw;
    if (fit > 0 && a.n_wg + a.n_dg + a.n_hw > fit && room >= tiles * 8 && a.n_wg <= 2 * room) {
      msplit = pick_ksplit_chunk(g.M, room / tiles, &chunk);      // effective split <= room / tiles
      a.wg.msplit = msplit; a.wg.mchunk = chunk;
      a.wg.out = msplit == 1 ? dwb : slabs;
      if (msplit_out) *msplit_out = msplit;
      a.wg_gz = msplit;
      a.n_wg = tiles * msplit;
    }
  }
  // stride-2 4x4 with 16 input channels (ImpalaCnnOpt conv2): sample-per-workgroup bf16x6 input gradient
  const bool s2c16 = tuning().bf16x6 && a.dg_direct == 0 && g.S == 2 && g.KH == 4 && g.KW == 4 && g.C == 16 && g.N == 32 &&
                     wsmall && (size_t)3 * (g.OHOW + 1) * 80 <= 8 * 1024 * 4;
  if (s2c16) {
    a.ddg.g = g; a.ddg.dy = dy; a.ddg.w = w; a.ddg.x = x_grad; a.ddg.dx = dx; a.ddg.act_prev = act_prev;
    a.ddg.mt = 0; a.ddg.ct = 0; a.ddg.deep = 0;
    a.dg_direct = 5;
    a.n_dg = B;
    if (tuning().bwd_deep_prefetch && 2 * B <= 256) { a.ddg.ct = 2; a.n_dg = 2 * B; }     // two workgroups per sample (four: no further gain)
  }
  // ... and, for large batches, the weight gradient in the same workgroups: 512 of them (two per CU), one slab each
  const bool s2fused = s2c16 && B >= 512 && slabs != nullptr && slab_cap >= 512 &&
                       (size_t)3 * (g.OHOW + 1) * 80 + (size_t)3 * (g.H * g.W + 1) * 32 <= 18 * 1024 * 4;
  if (s2fused) {
    a.n_dg = 512;                                    // each walks samples bid, bid + 512, ...
    a.n_wg = 0;
    a.wg.out = slabs;
    a.wg.msplit = a.n_dg;
    if (msplit_out) *msplit_out = a.n_dg;
  }
  const bool pad = is_padded(g);
  // staged-rows weight gradient (wgrad_rows_body) next to the all-classes input gradient: workgroup = (sample group,
  // kernel row), ceil(B / 64) samples per group so that KH * groups ~ one workgroup per CU, one slab per group
  bool wrows = false;
  if ((tuning().wgrad_rows == 1 || tuning().wgrad_rows == 2 || tuning().wgrad_rows == 3) && tuning().bf16x6 &&
      a.dg_direct == 2 && !pad && g.C == kWrC && g.N == kWrN && g.KW == kWrKW &&
      slabs != nullptr && wrows_smem_floats(g.OH, g.OW, g.W) <= kWrMaxSmemFloats &&
      g.OH * (g.W * kWrC / 4) <= kWrXQ * 256 && g.OHOW * (kWrN / 4) <= kWrDQ * 256) {
    const int per = (B + 63) / 64, groups = (B + per - 1) / per;
    if (groups <= slab_cap) {
      wrows = true;
      a.wg.mchunk = per;                  // samples per group
      a.wg.d_rowq = make_fastdiv((uint32_t)(g.W * kWrC / 4));
      a.wg.msplit = groups;
      a.wg.out = slabs;
      a.wg_gx = g.KH; a.wg_gy = 1; a.wg_gz = groups;
      a.n_wg = g.KH * groups;
      if (msplit_out) *msplit_out = groups;
    }
  }
  bool pf4_only = false;
  if (!wrows && tuning().wgrad_rows == 4 && tuning().bf16x6 && a.dg_direct == 2 && !pad &&
      (g.KH / g.S) * (g.KW / g.S) * (g.N >> 5) == 4) {
    const int tiles = a.wg_gx * a.wg_gy, room = 512 - a.n_dg - a.n_hw;
    if (room >= tiles * 8) {
      pf4_only = true;
      if (a.n_wg > room) {
        msplit = pick_ksplit_chunk(g.M, room / tiles, &chunk);
        a.wg.msplit = msplit; a.wg.mchunk = chunk;
        a.wg.out = msplit == 1 ? dwb : slabs;
        if (msplit_out) *msplit_out = msplit;
        a.wg_gz = msplit;
        a.n_wg = tiles * msplit;
      }
    }
  }
  const int total = a.n_wg + a.n_dg + a.n_hw;
#ifdef XT_TL_EXPERIMENT
  const bool exp_alone = wrows && tuning().wgrad_rows == 3;
#endif
  const bool dx6 = tuning().bf16x6 != 0 && a.dg_direct == 0;     // LDS-tiled input gradient on the bf16 matrix cores
  // Deep-prefetch instance of the generic LDS-tiled pair (Dense layers: S = 1, 1x1): every class reduction has at most 8
  // steps -> all input-gradient operands in flight, four register stages in the weight gradient, two workgroups per CU;
  // the weight-gradient split is cut so that the launch stays within 512 co-resident workgroups
  bool pf_generic = false;
  if (tuning().bwd_deep_prefetch && dx6 && !pad && !wsmall && !dsmall && g.S == 1 && g.KH == 1 && g.KW == 1 && g.N <= 256 &&
      a.dg_direct == 0) {
    // (the head weight-gradient blocks come last in block order and are short: they may spill into a second round)
    const int tiles = a.wg_gx * a.wg_gy, room = 512 - a.n_dg - a.n_hw;
    if (512 - a.n_dg >= tiles) {
      pf_generic = true;
      if (a.n_wg > room) {
        msplit = pick_ksplit_chunk(g.M, room / tiles > 1 ? room / tiles : 1, &chunk);
        a.wg.msplit = msplit; a.wg.mchunk = chunk;
        a.wg.out = msplit == 1 ? dwb : slabs;
        if (msplit_out) *msplit_out = msplit;
        a.wg_gz = msplit;
        a.n_wg = tiles * msplit;
      }
    }
  }
  const int total2 = a.n_wg + a.n_dg + a.n_hw;
  // a weight gradient written by ONE slab per tile is final: its blocks also leave their share of the squared global
  // norm, which saves the reduction launch a read of the whole tensor (PpoCnn's Dense layer: 6.4 of its 29 MB)
  const bool generic_wg = !wrows && !s2fused && !s2c16 && !halo_inst && a.dg_direct != 2;
  if (sq_partials && npre_out && generic_wg && a.wg.msplit == 1 && a.wg_gz == 1) {
    a.wg.sq_out = sq_partials; a.wg.sq_gx = a.wg_gx;
    *npre_out = a.wg_gx * a.wg_gy;
  }
#define XT_BWD2(WBI, WBJ, WWI, WWJ, DBI, DBJ, DWI, DWJ, X6V)                                                    \
  do {                                                                                                          \
    if (pad) hipLaunchKernelGGL((igemm_bwd_layer_kernel<WBI, WBJ, WWI, WWJ, true, DBI, DBJ, DWI, DWJ, 0, 0, X6V>), \
                                dim3(total), dim3(256), 0, st, a);                                              \
    else hipLaunchKernelGGL((igemm_bwd_layer_kernel<WBI, WBJ, WWI, WWJ, false, DBI, DBJ, DWI, DWJ, 0, 0, X6V>),  \
                            dim3(total), dim3(256), 0, st, a);                                                  \
  } while (0)
#define XT_BWD(WBI, WBJ, WWI, WWJ, DBI, DBJ, DWI, DWJ)                                                          \
  do {                                                                                                          \
    if (dx6) XT_BWD2(WBI, WBJ, WWI, WWJ, DBI, DBJ, DWI, DWJ, true);                                             \
    else XT_BWD2(WBI, WBJ, WWI, WWJ, DBI, DBJ, DWI, DWJ, false);                                                \
  } while (0)
  last_arith() = XT_ARITH_FP32;          // (register-direct input gradients and the x6 = 0 forms)
  if (dx6 || (tuning().bf16x6 && a.dg_direct == 2)) last_arith() = XT_ARITH_FP32_BF16X6;
  if (s2fused) {
    last_arith() = XT_ARITH_BF16X6;
    if (pad) hipLaunchKernelGGL((igemm_bwd_layer_kernel<128, 32, 4, 1, true, 128, 32, 4, 1, 0, 5>), dim3(total), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_bwd_layer_kernel<128, 32, 4, 1, false, 128, 32, 4, 1, 0, 5>), dim3(total), dim3(256), 0, st, a);
  } else if (s2c16) {
    last_arith() = XT_ARITH_FP32_BF16X6;
    if (pad) hipLaunchKernelGGL((igemm_bwd_layer_kernel<128, 32, 4, 1, true, 128, 32, 4, 1, 0, 3>), dim3(total), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_bwd_layer_kernel<128, 32, 4, 1, false, 128, 32, 4, 1, 0, 3>), dim3(total), dim3(256), 0, st, a);
  } else if (halo_inst) {
    const int hx6 = tuning().bf16x6;     // 0: fp32 MFMA (A/B)
    const int nsamp = 63 / (g.H * g.W) + 2;
    if (hx6 && (size_t)3 * (nsamp * g.OHOW + 1) * (g.N * 2 + 16) <= 44 * 1024) last_arith() = XT_ARITH_FP32_BF16X6;
    if (hx6 && (size_t)3 * (nsamp * g.OHOW + 1) * (g.N * 2 + 16) <= 44 * 1024) {
      if (tuning().dense_wgrad_x6 == 2)      // EXPERIMENT (off by default): conv3's weight gradient bf16x6 with one LDS stage
        hipLaunchKernelGGL((igemm_bwd_layer_kernel<64, 64, 2, 2, false, 128, 32, 4, 1, 0, 2, false, 0, 2>), dim3(total), dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL((igemm_bwd_layer_kernel<64, 64, 2, 2, false, 128, 32, 4, 1, 0, 2>), dim3(total), dim3(256), 0, st, a);
    }
    else
      hipLaunchKernelGGL((igemm_bwd_layer_kernel<64, 64, 2, 2, false, 128, 32, 4, 1, 0, 1>), dim3(total), dim3(256), 0, st, a);
  } else if (a.dg_direct == 2 && pf4_only) {
    // WROWS = 3: the LDS-tiled im2col weight gradient next to the all-taps-in-flight input gradient (250 VGPRs: two
    // workgroups per CU, the launch cut to 512 co-resident workgroups)
    hipLaunchKernelGGL((igemm_bwd_layer_kernel<128, 32, 4, 1, false, 128, 32, 4, 1, 2, 0, false, 3>), dim3(total), dim3(256), 0, st, a);
  } else if (a.dg_direct == 2 && wrows) {
    // (KH/S) * (KW/S) * (N/32) == 4 reduction steps: the input-gradient blocks keep all four taps' operands in flight
#ifdef XT_TL_EXPERIMENT
    if (exp_alone) a.dg_direct = 99;
#endif
    if ((g.KH / g.S) * (g.KW / g.S) * (g.N >> 5) == 4 && tuning().wgrad_rows != 2)
      hipLaunchKernelGGL((igemm_bwd_layer_kernel<128, 32, 4, 1, false, 128, 32, 4, 1, 2, 0, false, 2>), dim3(total), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((igemm_bwd_layer_kernel<128, 32, 4, 1, false, 128, 32, 4, 1, 2, 0, false, 1>), dim3(total), dim3(256), 0, st, a);
  } else if (a.dg_direct == 2) {
    XT_REQUIRE(wsmall && dsmall && !pad, "bwd_layer: the all-classes input gradient needs the small-tile configuration");
    const int x6 = tuning().bf16x6;      // 0: fp32 MFMA in the all-classes input gradient (A/B)
    if (x6) hipLaunchKernelGGL((igemm_bwd_layer_kernel<128, 32, 4, 1, false, 128, 32, 4, 1, 2>), dim3(total), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_bwd_layer_kernel<128, 32, 4, 1, false, 128, 32, 4, 1, 1>), dim3(total), dim3(256), 0, st, a);
  } else if (pf_generic) {
    if (tuning().dense_wgrad_x6) {        // (round 5) the Dense weight gradient on the bf16 matrix cores as well
      last_arith() = XT_ARITH_BF16X6;
      hipLaunchKernelGGL((igemm_bwd_layer_kernel<64, 64, 2, 2, false, 64, 64, 2, 2, 0, 0, true, 3, 1>), dim3(total2), dim3(256), 0, st, a);
    } else {
      hipLaunchKernelGGL((igemm_bwd_layer_kernel<64, 64, 2, 2, false, 64, 64, 2, 2, 0, 0, true, 3>), dim3(total2), dim3(256), 0, st, a);
    }
  } else if (wsmall && dsmall) XT_BWD(128, 32, 4, 1, 128, 32, 4, 1);
  else if (wsmall) XT_BWD(128, 32, 4, 1, 64, 64, 2, 2);
  else if (dsmall) XT_BWD(64, 64, 2, 2, 128, 32, 4, 1);
  else if (tuning().dense_wgrad_x6 && dx6 && !pad) {
    // the generic 64x64 pair (ImpalaCnnOpt's 11x11 "dense" conv) with the one-LDS-stage bf16x6 weight gradient (three
    // workgroups per CU as before): pong_impala_speedup 214.6 -> 212.3 us per 1000-frame train, breakout_impala unchanged
    last_arith() = XT_ARITH_BF16X6;
    hipLaunchKernelGGL((igemm_bwd_layer_kernel<64, 64, 2, 2, false, 64, 64, 2, 2, 0, 0, true, 0, 2>), dim3(total), dim3(256), 0, st, a);
  } else XT_BWD(64, 64, 2, 2, 64, 64, 2, 2);
#undef XT_BWD
#undef XT_BWD2
  XT_LAUNCH_CHECK();
  return 0;
}

}  // namespace xt

extern "C" {

int xt_layer_fwd(const xt_conv_geom* g, const xt_input_xform* xf, int32_t B, const void* in, const int32_t* idx,
                 const float* w, const float* bias, float* y, float* partial, int32_t ksplit, void* stream) {
  return xt::launch_fwd(g, xf, B, in, idx, w, bias, y, partial, ksplit, xt::as_stream(stream), nullptr, nullptr, nullptr);
}

int xt_layer_wgrad(const xt_conv_geom* g, const xt_input_xform* xf, int32_t B, const void* in, const int32_t* idx,
                   const float* dy, float* dwb, float* slabs, int32_t msplit, void* stream) {
  return xt::launch_wgrad(g, xf, B, in, idx, dy, dwb, slabs, msplit, xt::as_stream(stream), 1, nullptr, 0);
}

int xt_layer_dgrad(const xt_conv_geom* g, int32_t B, const float* dy, const float* w, const float* x,
                   int32_t act_prev, float* dx, void* stream) {
  return xt::launch_dgrad(g, B, dy, w, x, act_prev, dx, xt::as_stream(stream));
}

}  // extern "C"
