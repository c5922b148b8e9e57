// Device-side pieces of the register-direct implicit-GEMM family (see xt_direct.hip for the design notes) that
// are shared between translation units: the fused per-layer backward launch of xt_igemm.hip runs the direct
// input-gradient blocks next to its LDS-tiled weight-gradient blocks.
#pragma once
#include "xt_common.h"
#include "xt_igemm.h"
#include "xt_conv1_dev.h"

namespace xt {

// block id -> work item such that each XCD (block id mod 8, round-robin dispatch) gets a contiguous range
__device__ __forceinline__ uint32_t xcd_chunk(uint32_t bid, uint32_t nb) {
  const uint32_t q = nb >> 3, r = nb & 7u, x = bid & 7u, j = bid >> 3;
  return x * q + (x < r ? x : r) + j;
}

__device__ __forceinline__ float zsel(bool ok, float v) { return ok ? v : 0.f; }

// Buffer addressing (SRSRC descriptor in SGPRs + 32-bit lane byte offset + scalar byte offset): no 64-bit VALU
// address math, immediate-offset folding, and the hardware range check returns 0 for offsets >= num_bytes,
// which is how padding taps and tail rows are zero-filled for free (kOob).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kOob = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

// ------------------------------------------------------------------ input gradient
// Static-NW combine of the per-wave partial tiles, emitted as 16-byte row pieces.  red[q][r][kl*32 + il] holds
// element (row(r, kl), column il) of wave q's partial tile (r = tile*16 + reg), so four consecutive columns of a
// row are contiguous in LDS.  Float4 slot e (of R*16): columns 4*(e & 7).., kl = (e >> 3) & 1, r = e >> 4; lane
// `lane` of wave w owns slots e = w*64 + lane + j*NW*64 -> eight consecutive lanes cover one 128-byte row of a
// 32-column tile.  Fixed summation order q = 0..NW-1.
template <int R, int NW>
struct Slot4 { int c4, kl, r; };
template <int R, int NW>
__device__ __forceinline__ Slot4<R, NW> slot4(int w, int lane, int j) {
  const int e = w * 64 + lane + j * NW * 64;
  return Slot4<R, NW>{(e & 7) * 4, (e >> 3) & 1, e >> 4};
}
template <int R, int NW, typename F>
__device__ __forceinline__ void combine_emit_static4(float* red, const float (&flat)[R], int w, int lane, F emit) {
#pragma unroll
  for (int r = 0; r < R; ++r) red[(w * R + r) * 64 + lane] = flat[r];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < R * 16 / (NW * 64); ++j) {
    const Slot4<R, NW> sl = slot4<R, NW>(w, lane, j);
    float4 v = *reinterpret_cast<const float4*>(&red[sl.r * 64 + sl.kl * 32 + sl.c4]);
#pragma unroll
    for (int q = 1; q < NW; ++q) {
      const float4 u = *reinterpret_cast<const float4*>(&red[(q * R + sl.r) * 64 + sl.kl * 32 + sl.c4]);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    emit(j, v);
  }
}

struct DDgradArgs {
  Geom g;
  const float* dy;
  const float* w;
  const float* x;     // producer's post-activation output [B,H,W,C]
  float* dx;
  int act_prev;
  int mt, ct;         // pixel tiles per stride-parity class (upper bound), channel tiles
  int deep;           // halo form: every weight stage of a wave's slice in flight (tuning.bwd_deep_prefetch)
};

// bid / nblocks: this block's index among the launch's input-gradient blocks (the fused per-layer backward
// launch of xt_igemm.hip runs them next to the weight-gradient blocks); red: >= NW*TI*TJ*1024 floats of LDS
// (also for NW = 1: the tile is transposed through it).
template <int TI, int TJ, int NW>
__device__ __forceinline__ void direct_dgrad_body(const DDgradArgs& p, uint32_t bid, uint32_t nblocks, float* red) {
  const Geom& g = p.g;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int il = lane & 31, kl = lane >> 5;
  XT_TL(0);
  XT_TL_ROLE(80);
  // tile-major, class-minor: the S*S classes of one pixel region read the same dY lines
  uint32_t lin = xcd_chunk(bid, nblocks);
  const int nclass = g.S * g.S;
  const int cls = (int)(lin % (uint32_t)nclass);
  lin /= (uint32_t)nclass;
  const int tc = (int)(lin % (uint32_t)p.ct), tm = (int)(lin / (uint32_t)p.ct);
  const int ry = cls / g.S, rx = cls - ry * g.S;
  const int cy0 = ((ry - g.PT) % g.S + g.S) % g.S, cx0 = ((rx - g.PL) % g.S + g.S) % g.S;
  const int HC = cy0 < g.H ? (g.H - cy0 + g.S - 1) / g.S : 0;
  const int WC = cx0 < g.W ? (g.W - cx0 + g.S - 1) / g.S : 0;
  const int Mc = g.B * HC * WC;
  const int i0 = tm * (32 * TI), c0 = tc * (32 * TJ);
  if (i0 >= Mc) return;
  const int JY = ry < g.KH ? (g.KH - ry + g.S - 1) / g.S : 0;
  const int JX = rx < g.KW ? (g.KW - rx + g.S - 1) / g.S : 0;
  const int nps = g.N >> 5;                        // 32-deep steps per tap
  const int nsteps = JY * JX * nps;
  const int qy0 = (cy0 + g.PT) / g.S, qx0 = (cx0 + g.PL) / g.S;
  const int per = (nsteps + NW - 1) / NW;
  const int s0 = w * per, s1 = min(nsteps, s0 + per);

  int dybase[TI], qy[TI], qx[TI], outoff[TI];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
    const int mraw = i0 + 32 * ti + il;
    const int mc = min(mraw, Mc - 1);
    const int b = mc / (HC * WC), rem = mc - b * (HC * WC);
    const int ty = rem / WC, tx = rem - ty * WC;
    qy[ti] = qy0 + ty; qx[ti] = qx0 + tx;
    dybase[ti] = ((b * g.OH + qy[ti]) * g.OW + qx[ti]) * g.N;
    outoff[ti] = mraw < Mc ? ((b * g.H + cy0 + g.S * ty) * g.W + cx0 + g.S * tx) * g.C : -1;
  }
  const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(p.dy, (uint32_t)g.M * (uint32_t)g.N * 4u);
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (uint32_t)g.K * (uint32_t)g.N * 4u);
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, (uint32_t)g.B * (uint32_t)g.HWC * 4u);

  // producer activations of the 16-byte pieces this lane will emit: loaded before the reduction loop
  constexpr int R = TI * TJ * 16, RJ = R * 16 / (NW * 64);
  static_assert(R * 16 % (NW * 64) == 0, "direct dgrad: tile does not split evenly over the waves");
  float4 xv[RJ];
  int eoff[RJ];
#pragma unroll
  for (int j = 0; j < RJ; ++j) {
    const Slot4<R, NW> sl = slot4<R, NW>(w, lane, j);
    const int t = sl.r >> 4, rr = sl.r & 15;
    const int ti = t / TJ, tj = t - ti * TJ;
    const int row = (rr & 3) + 8 * (rr >> 2) + 4 * sl.kl;
    int o = 0;
#pragma unroll
    for (int q = 0; q < TI; ++q) { const int oq = __shfl(outoff[q], row, 64); if (q == ti) o = oq; }
    eoff[j] = o >= 0 ? o + c0 + 32 * tj + sl.c4 : -1;
    xv[j] = buf_load4(rs_x, o >= 0 ? (uint32_t)eoff[j] * 4u : kOob, 0u);
  }

  const uint32_t wvoff = (uint32_t)((c0 + il) * g.N + 16 * kl) * 4u;
  struct Stage { float4 a[TI][4]; float4 b[TJ][4]; };
  auto load = [&](Stage& S_, int s, bool live) {
    const int tap = s / nps;                       // uniform
    const int jy = JX > 0 ? tap / JX : 0, jx = tap - jy * JX;
    const int nofs = (s - tap * nps) * 32;
    const int tapoff = (jy * g.OW + jx) * g.N - nofs - 16 * kl;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      const bool ok = live && ((unsigned)(qy[ti] - jy) < (unsigned)g.OH) && ((unsigned)(qx[ti] - jx) < (unsigned)g.OW);
      const uint32_t off = ok ? (uint32_t)(dybase[ti] - tapoff) * 4u : kOob;
#pragma unroll
      for (int q = 0; q < 4; ++q) S_.a[ti][q] = buf_load4(rs_dy, off + 16u * q, 0u);
    }
    // dead stage: out of range through the LANE offset (the range check covers voffset + immediate, not soffset)
    const uint32_t ws = live ? (uint32_t)((((ry + g.S * jy) * g.KW + rx + g.S * jx) * g.C) * g.N + nofs) * 4u : 0u;
    const uint32_t vo = live ? wvoff : kOob;
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int q = 0; q < 4; ++q) S_.b[tj][q] = buf_load4(rs_w, vo + (uint32_t)(32 * tj * g.N) * 4u + 16u * q, ws);
  };
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
  auto pick = [](const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; };
  auto compute = [&](const Stage& S_) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(pick(S_.a[ti][kk >> 2], kk & 3), pick(S_.b[tj][kk >> 2], kk & 3),
                                                             acc[ti][tj], 0, 0, 0);
  };
  Stage R0, R1;
  load(R0, s0, s0 < s1);
  load(R1, s0 + 1, s0 + 1 < s1);
  XT_TL(1);
  for (int s = s0; s < s1; s += 2) {
    compute(R0);
    load(R0, s + 2, s + 2 < s1);
    compute(R1);
    load(R1, s + 3, s + 3 < s1);
  }
  XT_TL(3);
  float flat[R];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) flat[(ti * TJ + tj) * 16 + r] = acc[ti][tj][r];
  combine_emit_static4<R, NW>(red, flat, w, lane, [&](int j, float4 v) {
    if (eoff[j] >= 0) {
      v.x *= act_grad(xv[j].x, p.act_prev); v.y *= act_grad(xv[j].y, p.act_prev);
      v.z *= act_grad(xv[j].z, p.act_prev); v.w *= act_grad(xv[j].w, p.act_prev);
      store4_wt(p.dx, (size_t)eoff[j], v);
    }
  });
  XT_TL(4);
  XT_TL_DRAIN(5);
}

// ------------------------------------------------------------------ stride-1 input gradient, dY halo in LDS
// Same tile / wave / combine structure as direct_dgrad_body<2, 1, 4> (64 positions x 32 channels per workgroup,
// the four waves split the KH*KW*N reduction, weights straight from L2 into registers), but the dY operand is not
// gathered from global memory per tap: the KH*KW taps of a position re-read the same few dY rows, every wave of
// the workgroup walks other taps of the SAME rows, and with two or three co-resident workgroups that working set
// (about 21 KB each) falls out of the 32 KB L1 -- the register-direct form got slower with more co-residency
// (late-round finding 4).  A 64-position tile of a small stride-1 map touches at most `nsamp` samples;
// their whole dY ([OH*OW][N] each, one contiguous block) is staged once into LDS with rows padded to N + 4 floats
// (16-byte reads of 32 consecutive rows then hit all banks), plus one zero row that out-of-range taps point to.
// Requires S == 1; smem >= max(staged rows + 1 zero row, 4 waves x 2 tiles x 4 KB for the combine).
// SPLIT = true: "bf16x6" (see split3_store in xt_conv1_dev.h): dY is split into three bf16 planes when it is
// staged -- [plane][row][N bf16 + 16 B pad], one ds_read_b128 per MFMA operand -- and the weights, which go
// global -> VGPR, are split in registers (8 values per 16-deep chunk).  Within a 32-deep step lane (il, kl) owns the
// reduction indices 16*chunk + 8*kl + j of both operands.
template <bool SPLIT>
__device__ __forceinline__ void halo_dgrad_body(const DDgradArgs& p, uint32_t bid, uint32_t nblocks, float* smem) {
  constexpr int TI = 2, NW = 4, R = TI * 16, RJ = R * 16 / (NW * 64);
  const Geom& g = p.g;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int il = lane & 31, kl = lane >> 5;
  XT_TL(0);
  XT_TL_ROLE(80);
  const uint32_t lin = xcd_chunk(bid, nblocks);
  const int tc = (int)(lin % (uint32_t)p.ct), tm = (int)(lin / (uint32_t)p.ct);
  const int HW = g.H * g.W, Mc = g.B * HW;
  const int i0 = tm * (32 * TI), c0 = tc * 32;
  if (i0 >= Mc) return;                                   // block-uniform
  const int sfirst = i0 / HW, slast = min(i0 + 32 * TI - 1, Mc - 1) / HW;
  const int RS = g.N + 4;                                 // fp32 form: LDS row stride in floats
  const int RB = g.N * 2 + 16;                            // split form: bytes per row of one bf16 plane
  const int nrows = (slast - sfirst + 1) * g.OHOW;        // staged dY rows; row `nrows` is the zero row
  uint8_t* sb = reinterpret_cast<uint8_t*>(smem);
  const int PS = (nrows + 1) * RB;                        // split form: bytes per plane
  {
    const float4* src = reinterpret_cast<const float4*>(p.dy + (size_t)sfirst * g.OHOW * g.N);
    const int n4row = g.N >> 2, total4 = nrows * n4row;
    for (int base = 0; base < total4; base += 256 * 8) {
      float4 v[8];
      int dst[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        int e = base + t + 256 * q;
        e = e < total4 ? e : 0;                            // past the end: re-copy element 0 (unconditional LDS write)
        const int row = e / n4row;
        v[q] = src[e];
        dst[q] = SPLIT ? row * RB + (e - row * n4row) * 8 : row * RS + (e - row * n4row) * 4;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if constexpr (SPLIT) split3_store(sb + dst[q], PS, v[q]);
        else *reinterpret_cast<float4*>(&smem[dst[q]]) = v[q];
      }
    }
    if constexpr (SPLIT) {
      for (int e = t; e < 3 * (RB >> 2); e += 256) {
        const int pl = e / (RB >> 2);
        reinterpret_cast<uint32_t*>(sb + pl * PS + nrows * RB)[e - pl * (RB >> 2)] = 0u;
      }
    } else {
      for (int e = t; e < RS; e += 256) smem[nrows * RS + e] = 0.f;
    }
  }
  const int nps = g.N >> 5;
  const int nsteps = g.KH * g.KW * nps;
  const int per = (nsteps + NW - 1) / NW;
  const int s0 = w * per, s1 = min(nsteps, s0 + per);

  int r0[TI], qy[TI], qx[TI], outoff[TI];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
    const int mraw = i0 + 32 * ti + il;
    const int mc = min(mraw, Mc - 1);
    const int b = mc / HW, rem = mc - b * HW;
    const int y = rem / g.W, x = rem - y * g.W;
    qy[ti] = y + g.PT; qx[ti] = x + g.PL;
    r0[ti] = (b - sfirst) * g.OHOW + qy[ti] * g.OW + qx[ti];
    outoff[ti] = mraw < Mc ? mc * g.C : -1;
  }
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (uint32_t)g.K * (uint32_t)g.N * 4u);
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, (uint32_t)g.B * (uint32_t)g.HWC * 4u);
  float4 xv[RJ];
  int eoff[RJ];
#pragma unroll
  for (int j = 0; j < RJ; ++j) {
    const Slot4<R, NW> sl = slot4<R, NW>(w, lane, j);
    const int ti = sl.r >> 4, rr = sl.r & 15;
    const int row = (rr & 3) + 8 * (rr >> 2) + 4 * sl.kl;
    int o = 0;
#pragma unroll
    for (int q = 0; q < TI; ++q) { const int oq = __shfl(outoff[q], row, 64); if (q == ti) o = oq; }
    eoff[j] = o >= 0 ? o + c0 + sl.c4 : -1;
    xv[j] = buf_load4(rs_x, o >= 0 ? (uint32_t)eoff[j] * 4u : kOob, 0u);
  }
  const uint32_t wvoff = (uint32_t)((c0 + il) * g.N + (SPLIT ? 8 : 16) * kl) * 4u;
  struct StageA { float4 a[TI][4]; };
  struct StageB { float4 b[4]; };
  auto loadB = [&](StageB& S_, int s, bool live) {
    const int tap = s / nps;
    const uint32_t ws = live ? (uint32_t)((tap * g.C) * g.N + (s - tap * nps) * 32) * 4u : 0u;
    const uint32_t vo = live ? wvoff : kOob;
#pragma unroll
    for (int q = 0; q < 4; ++q) S_.b[q] = buf_load4(rs_w, vo + (SPLIT ? 64u * (q >> 1) + 16u * (q & 1) : 16u * q), ws);
  };
  auto loadA = [&](StageA& S_, int s, bool live) {
    const int tap = s / nps;
    const int jy = tap / g.KW, jx = tap - jy * g.KW;
    const int nofs = (s - tap * nps) * 32 + 16 * kl;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      const bool ok = live && ((unsigned)(qy[ti] - jy) < (unsigned)g.OH) && ((unsigned)(qx[ti] - jx) < (unsigned)g.OW);
      const int row = ok ? r0[ti] - (jy * g.OW + jx) : nrows;
      const float* ap = smem + row * RS + nofs;
#pragma unroll
      for (int q = 0; q < 4; ++q) S_.a[ti][q] = *reinterpret_cast<const float4*>(ap + 4 * q);
    }
  };
  f32x16 acc[TI];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;
  auto pick = [](const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; };
  auto compute = [&](const StageA& A_, const StageB& B_) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
        acc[ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(pick(A_.a[ti][kk >> 2], kk & 3), pick(B_.b[kk >> 2], kk & 3), acc[ti], 0, 0, 0);
  };
  // split form: one step = 2 chunks x (TI x 3 plane reads from LDS, weight chunk split in registers, TI x 6 MFMAs)
  auto step_split = [&](const StageB& B_, int s, bool live) {
    const int tap = s / nps;
    const int jy = tap / g.KW, jx = tap - jy * g.KW;
    const int nofs = (s - tap * nps) * 32;
    int rowb[TI];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      const bool ok = live && ((unsigned)(qy[ti] - jy) < (unsigned)g.OH) && ((unsigned)(qx[ti] - jx) < (unsigned)g.OW);
      rowb[ti] = (ok ? r0[ti] - (jy * g.OW + jx) : nrows) * RB + (nofs + 8 * kl) * 2;
    }
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      bf16x8 a[TI][3], b[3];
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[ti][pl] = *reinterpret_cast<const bf16x8*>(sb + pl * PS + rowb[ti] + ch * 32);
      split3_regs(B_.b[2 * ch], B_.b[2 * ch + 1], b);
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) acc[ti] = mfma_bf16x6(a[ti], b, acc[ti]);
    }
  };
  StageB B0, B1;
  loadB(B0, s0, s0 < s1);
  loadB(B1, s0 + 1, s0 + 1 < s1);
  __syncthreads();                                        // staged dY visible
  XT_TL(1);
  if constexpr (SPLIT) {
    // three weight stages in flight (the split form has the registers for it: 120 -> 136 VGPRs): a wave's whole slice
    // is at most ceil(KH*KW*N/32 / 4) steps, so most of its weight loads are issued before the first MFMA
    StageB B2;
    loadB(B2, s0 + 2, s0 + 2 < s1);
    if (per <= 5 && p.deep) {
      // (round 3) a wave's slice is at most five steps: ALL of its weight loads in flight before the first MFMA, straight-line
      StageB B3, B4;
      loadB(B3, s0 + 3, s0 + 3 < s1);
      loadB(B4, s0 + 4, s0 + 4 < s1);
      step_split(B0, s0, s0 < s1);
      step_split(B1, s0 + 1, s0 + 1 < s1);
      step_split(B2, s0 + 2, s0 + 2 < s1);
      step_split(B3, s0 + 3, s0 + 3 < s1);
      step_split(B4, s0 + 4, s0 + 4 < s1);
    } else
    for (int s = s0; s < s1; s += 3) {
      step_split(B0, s, true);
      loadB(B0, s + 3, s + 3 < s1);
      step_split(B1, s + 1, s + 1 < s1);
      loadB(B1, s + 4, s + 4 < s1);
      step_split(B2, s + 2, s + 2 < s1);
      loadB(B2, s + 5, s + 5 < s1);
    }
  } else {
    StageA A0, A1;
    loadA(A0, s0, s0 < s1);
    for (int s = s0; s < s1; s += 2) {
      loadA(A1, s + 1, s + 1 < s1);
      compute(A0, B0);
      loadB(B0, s + 2, s + 2 < s1);
      loadA(A0, s + 2, s + 2 < s1);
      compute(A1, B1);
      loadB(B1, s + 3, s + 3 < s1);
    }
  }
  XT_TL(3);
  float flat[R];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) flat[ti * 16 + r] = acc[ti][r];
  __syncthreads();                                        // every wave is done reading the staged rows: reuse as `red`
  combine_emit_static4<R, NW>(smem, flat, w, lane, [&](int j, float4 v) {
    if (eoff[j] >= 0) {
      v.x *= act_grad(xv[j].x, p.act_prev); v.y *= act_grad(xv[j].y, p.act_prev);
      v.z *= act_grad(xv[j].z, p.act_prev); v.w *= act_grad(xv[j].w, p.act_prev);
      store4_wt(p.dx, (size_t)eoff[j], v);
    }
  });
  XT_TL(4);
  XT_TL_DRAIN(5);
}

// ------------------------------------------------------------------ stride-2 4x4 input gradient, 16 input channels
// ImpalaCnnOpt's second conv (21x21x16 -> 11x11x32, 4x4 / 2, SAME; xt/model/atari_model.py:8-17): the generic
// per-class kernel gives its 16 input channels a 32-column MFMA tile (half empty), decodes 128 class positions per
// workgroup for four 32-deep steps and re-gathers dY per class from global memory -- 61 of pong_impala_speedup's 270 us.
// Here ONE workgroup owns one SAMPLE: its dY ([OH*OW][32], 15.5 KB) is staged once into LDS as bf16 planes (+ a zero
// row that padding taps point to), and wave w computes parity class (w >> 1, w & 1) of the sample's input pixels with
// v_mfma_f32_16x16x32_bf16 -- 16 positions x 16 channels per accumulator, one 32-deep MFMA slab = one tap (N = 32) --
// in the bf16x6 form; the class's four weight taps live in registers (split once).  Requires S = 2, KH = KW = 4,
// C = 16 or 32 (NC channel tiles), N = 32; smem >= 3 * (OH*OW + 1) * 80 bytes.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16_bf16x6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
  return acc;
}

template <int NC>      // NC = C / 16 channel tiles (1: ImpalaCnnOpt conv2; 2 = PpoCnn conv2 measured slower than the
                       // all-classes tiles of igemm_dgrad4_body and is not dispatched)
__device__ __forceinline__ void s2c16_dgrad_body(const DDgradArgs& p, uint32_t bid, float* smem) {
  constexpr int C = 16 * NC;
  const Geom& g = p.g;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int c = lane & 15, g4 = lane >> 4;               // operand role: column (channel) / row c, k group g4
  XT_TL(0);
  XT_TL_ROLE(80);
  // p.ct (> 1): that many workgroups share a sample, each taking every p.ct-th 16-position tile of every class -- small
  // batches (128 frames = 128 workgroups) left half of the CUs without an input-gradient block
  const int nsplit = p.ct > 1 ? p.ct : 1;
  const int b = (int)bid / nsplit, part = (int)bid - b * nsplit;
  constexpr int RB = 32 * 2 + 16;                         // bytes per plane row (N = 32 bf16 + pad)
  const int nrows = g.OHOW;
  const int PS = (nrows + 1) * RB;
  uint8_t* sb = reinterpret_cast<uint8_t*>(smem);
  {                                                       // stage + split this sample's dY
    const float4* src = reinterpret_cast<const float4*>(p.dy + (size_t)b * g.OHOW * 32);
    const int total4 = nrows * 8;
    for (int base = 0; base < total4; base += 256 * 4) {
      float4 v[4];
      int dst[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int e = base + t + 256 * q;
        e = e < total4 ? e : 0;
        v[q] = src[e];
        dst[q] = (e >> 3) * RB + (e & 7) * 8;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) split3_store(sb + dst[q], PS, v[q]);
    }
    for (int e = t; e < 3 * (RB >> 2); e += 256) {
      const int pl = e / (RB >> 2);
      reinterpret_cast<uint32_t*>(sb + pl * PS + nrows * RB)[e - pl * (RB >> 2)] = 0u;
    }
  }
  // this wave's parity class
  const int ry = w >> 1, rx = w & 1;
  const int cy0 = ((ry - g.PT) % 2 + 2) % 2, cx0 = ((rx - g.PL) % 2 + 2) % 2;
  const int HC = cy0 < g.H ? (g.H - cy0 + 1) / 2 : 0, WC = cx0 < g.W ? (g.W - cx0 + 1) / 2 : 0;
  const int Mc = HC * WC;
  const int qy0 = (cy0 + g.PT) / 2, qx0 = (cx0 + g.PL) / 2;
  // the class's four taps (ky, kx) = (ry + 2 jy, rx + 2 jx): B operand of lane (channel c [+ 16 per tile], k group g4)
  // = 8 consecutive n
  bf16x8 wreg[4][NC][3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ky = ry + 2 * (j >> 1), kx = rx + 2 * (j & 1);
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      const float* wp = p.w + (size_t)((ky * 4 + kx) * C + 16 * ct + c) * 32 + 8 * g4;
      split3_regs(*reinterpret_cast<const float4*>(wp), *reinterpret_cast<const float4*>(wp + 4), wreg[j][ct]);
    }
  }
  __syncthreads();
  XT_TL(1);
  const size_t xbase = (size_t)b * g.H * g.W * C;
  // The producer activations of a tile's outputs (for act') are requested ONE TILE AHEAD (clamped, unconditional): read
  // inside the epilogue they exposed a global round trip per 16-position tile -- 10 of the 12 us a sample's workgroup
  // lived at 128 frames (timeline) -- with only ~24 MFMAs to hide behind.
  auto tile_pix = [&](int sub) {
    const int pos = min(sub + c, Mc - 1);
    const int ty = pos / WC, tx = pos - ty * WC;
    return (cy0 + 2 * ty) * g.W + cx0 + 2 * tx;
  };
  auto load_x = [&](int sub, float (&xv)[NC][4]) {
    const int pixn = tile_pix(sub < Mc ? sub : 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rpix = __shfl(pixn, 4 * g4 + i, 64);
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) xv[ct][i] = p.x[xbase + (size_t)rpix * C + 16 * ct + c];
    }
  };
  float xcur[NC][4], xnext[NC][4];
  load_x(16 * part, xcur);
  for (int sub = 16 * part; sub < Mc; sub += 16 * nsplit) {
    load_x(sub + 16 * nsplit, xnext);
    const int pos = min(sub + c, Mc - 1);                 // A-operand row of this lane: class position sub + c
    const int ty = pos / WC, tx = pos - ty * WC;
    const int pix = (cy0 + 2 * ty) * g.W + cx0 + 2 * tx;  // input pixel of that row
    f32x4 acc[NC];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int oy = ty - (j >> 1) + qy0, ox = tx - (j & 1) + qx0;
      const bool ok = ((unsigned)oy < (unsigned)g.OH) && ((unsigned)ox < (unsigned)g.OW);
      const uint8_t* ap = sb + (ok ? oy * g.OW + ox : nrows) * RB + 16 * g4;
      bf16x8 a[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const bf16x8*>(ap + pl * PS);
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) acc[ct] = mfma16_bf16x6(a, wreg[j][ct], acc[ct]);
    }
    // accumulator element i of lane (c, g4) = (row 4 g4 + i, channel 16 ct + c): that row's pixel comes from lane 4 g4 + i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * g4 + i;
      const int rpix = __shfl(pix, row, 64);
      if (sub + row < Mc) {
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
          const size_t off = xbase + (size_t)rpix * C + 16 * ct + c;
          p.dx[off] = acc[ct][i] * act_grad(xcur[ct][i], p.act_prev);
        }
      }
    }
#pragma unroll
    for (int ct = 0; ct < NC; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) xcur[ct][i] = xnext[ct][i];
  }
  XT_TL(4);
  XT_TL_DRAIN(5);
}

// ------------------------------------------------------------------ whole backward of the same layer, per sample
// ImpalaCnnOpt conv2 again (C = 16, N = 32, 4x4 / 2), for LARGE batches: input gradient AND weight gradient of a sample
// from ONE staging of its dY and of its input activation map -- the separate weight-gradient blocks re-read both through
// im2col gathers (31 us at 1000 frames on top of the input gradient).  A workgroup walks samples bid, bid + nblk, ...;
// per sample:
//   stage dY ([OH*OW][32]) and x ([H*W][16]) as bf16 planes (+ a zero row / zero pixel), barrier;
//   input gradient exactly as s2c16_dgrad_body (class per wave);
//   weight gradient dW[(tap, c)][n] += sum_pos x[pixel(pos, tap)][c] dY[pos][n] on 16x16x32 tiles -- a tap's 16 channels
//   are one 16-row tile, 32 positions one MFMA slab.  Both operands keep their NATURAL LDS layout ([pixel][c], [pos][n])
//   and are gathered by ds_read_b64_tr_b16: lanes 4j..4j+3 of a 16-lane group address position j's 16 channels /
//   columns -- the pixel of a tap is a per-lane address, so the im2col gather costs nothing; wave w owns kernel row
//   ky = w (4 taps) x both 16-column halves = 8 accumulators that live across the workgroup's samples.
// One slab [(K+1)*32] per workgroup; TWO workgroups per CU (72 KB of LDS each) so that one's staging latency hides
// behind the other's MFMAs: with one per CU (256 slabs) every CU walked its samples strictly in sequence and the launch
// was slower than the split form (58.5 vs 56.1 us at 1000 frames).  smem >= 3*(OH*OW+1)*80 + 3*(H*W+1)*32 bytes.
__device__ __forceinline__ bf16x8 lds_tr16x2(const uint8_t* p0, const uint8_t* p1) {
  typedef short i16x4 __attribute__((ext_vector_type(4)));
  union { i16x4 h[2]; bf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p0));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p1));
  return u.v;
}

__device__ __forceinline__ void s2c16_bwd_body(const DDgradArgs& p, float* slabs, int nblk, uint32_t bid, float* smem) {
  const Geom& g = p.g;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int c = lane & 15, g4 = lane >> 4;
  XT_TL(0);
  XT_TL_ROLE(80);
  constexpr int RB = 32 * 2 + 16, PB = 32;                // bytes per dY plane row / per image plane pixel (16 bf16)
  const int nrows = g.OHOW, HW = g.H * g.W;
  const int PS = (nrows + 1) * RB, IPS = (HW + 1) * PB;
  uint8_t* sb = reinterpret_cast<uint8_t*>(smem);         // dY planes
  uint8_t* img = sb + 3 * PS;                             // x planes
  // input-gradient role: parity class of this wave, its four taps in registers
  const int ry = w >> 1, rx = w & 1;
  const int cy0 = ((ry - g.PT) % 2 + 2) % 2, cx0 = ((rx - g.PL) % 2 + 2) % 2;
  const int HC = cy0 < g.H ? (g.H - cy0 + 1) / 2 : 0, WC = cx0 < g.W ? (g.W - cx0 + 1) / 2 : 0;
  const int Mc = HC * WC;
  const int qy0 = (cy0 + g.PT) / 2, qx0 = (cx0 + g.PL) / 2;
  bf16x8 wreg[4][3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ky = ry + 2 * (j >> 1), kx = rx + 2 * (j & 1);
    const float* wp = p.w + (size_t)((ky * 4 + kx) * 16 + c) * 32 + 8 * g4;
    split3_regs(*reinterpret_cast<const float4*>(wp), *reinterpret_cast<const float4*>(wp + 4), wreg[j]);
  }
  // weight-gradient role: kernel row ky = w, taps kx = 0..3, both column halves
  f32x4 dw[4][2];
#pragma unroll
  for (int kx = 0; kx < 4; ++kx)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) dw[kx][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int jrow = c >> 2, mq = c & 3;                    // tr-read role: position row j of the 4-row read, 8-byte quarter
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};                   // bias gradient: columns 4 * (t & 7) .. + 3 of this thread's rows

  for (int b = (int)bid; b < g.B; b += nblk) {
    {   // ---- stage + split dY and x of this sample
      const float4* src = reinterpret_cast<const float4*>(p.dy + (size_t)b * g.OHOW * 32);
      const int total4 = nrows * 8;
      for (int base = 0; base < total4; base += 256 * 4) {
        float4 v[4];
        int dst[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = base + t + 256 * q;
          const bool ok = e < total4;
          v[q] = src[ok ? e : 0];
          if (!ok) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
          dst[q] = ok ? (e >> 3) * RB + (e & 7) * 8 : nrows * RB + (t & 7) * 8;     // past the end: zeros into the zero row
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bsum[0] += v[q].x; bsum[1] += v[q].y; bsum[2] += v[q].z; bsum[3] += v[q].w;
          split3_store(sb + dst[q], PS, v[q]);
        }
      }
      const float4* xs = reinterpret_cast<const float4*>(p.x + (size_t)b * HW * 16);
      const int xt4 = HW * 4;
      for (int base = 0; base < xt4; base += 256 * 4) {
        float4 v[4];
        int dst[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = base + t + 256 * q;
          const bool ok = e < xt4;
          v[q] = xs[ok ? e : 0];
          if (!ok) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
          dst[q] = ok ? (e >> 2) * PB + (e & 3) * 8 : HW * PB + (t & 3) * 8;         // ... zeros into the zero pixel
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) split3_store(img + dst[q], IPS, v[q]);
      }
      if (t < 3 * (RB >> 2)) {                             // zero row of dY, zero pixel of x (every sample: cheap)
        const int pl = t / (RB >> 2);
        reinterpret_cast<uint32_t*>(sb + pl * PS + nrows * RB)[t - pl * (RB >> 2)] = 0u;
      }
      if (t < 3 * (PB >> 2)) {
        const int pl = t / (PB >> 2);
        reinterpret_cast<uint32_t*>(img + pl * IPS + HW * PB)[t - pl * (PB >> 2)] = 0u;
      }
    }
    __syncthreads();
    // ---- input gradient of this wave's parity class (s2c16_dgrad_body)
    const size_t xbase = (size_t)b * HW * 16;
    auto load_x = [&](int sub, float (&xv)[4]) {          // producer activations of tile `sub`, requested a tile ahead
      const int posn = min((sub < Mc ? sub : 0) + c, Mc - 1);
      const int tyn = posn / WC, txn = posn - tyn * WC;
      const int pixn = (cy0 + 2 * tyn) * g.W + cx0 + 2 * txn;
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[i] = p.x[xbase + (size_t)__shfl(pixn, 4 * g4 + i, 64) * 16 + c];
    };
    float xcur[4], xnext[4];
    load_x(0, xcur);
    for (int sub = 0; sub < Mc; sub += 16) {
      load_x(sub + 16, xnext);
      const int pos = min(sub + c, Mc - 1);
      const int ty = pos / WC, tx = pos - ty * WC;
      const int pix = (cy0 + 2 * ty) * g.W + cx0 + 2 * tx;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int oy = ty - (j >> 1) + qy0, ox = tx - (j & 1) + qx0;
        const bool ok = ((unsigned)oy < (unsigned)g.OH) && ((unsigned)ox < (unsigned)g.OW);
        const uint8_t* ap = sb + (ok ? oy * g.OW + ox : nrows) * RB + 16 * g4;
        bf16x8 a[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const bf16x8*>(ap + pl * PS);
        acc = mfma16_bf16x6(a, wreg[j], acc);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 4 * g4 + i;
        const int rpix = __shfl(pix, row, 64);
        if (sub + row < Mc) {
          const size_t off = xbase + (size_t)rpix * 16 + c;
          p.dx[off] = acc[i] * act_grad(xcur[i], p.act_prev);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) xcur[i] = xnext[i];
    }
    // ---- weight gradient: kernel row ky = w
    for (int s0 = 0; s0 < nrows; s0 += 32) {
      int prow[2], piy[2], pix0[2];                       // the two 4-position rows this lane addresses in the slab
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int pp = s0 + 8 * g4 + 4 * r + jrow;
        const bool live = pp < nrows;
        const int pc = live ? pp : 0;
        const int oy = pc / g.OW, ox = pc - oy * g.OW;
        prow[r] = live ? pp : nrows;                        // dY row (zero row past the end)
        piy[r] = live ? 2 * oy - g.PT + w : -(1 << 20);     // input row of tap row ky = w
        pix0[r] = 2 * ox - g.PL;                            // input column of tap kx = 0
      }
      bf16x8 bop[2][3];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          bop[nt][pl] = lds_tr16x2(sb + pl * PS + prow[0] * RB + (nt * 16 + 4 * mq) * 2,
                                   sb + pl * PS + prow[1] * RB + (nt * 16 + 4 * mq) * 2);
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        int pidx[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int x = pix0[r] + kx;
          const bool ok = ((unsigned)piy[r] < (unsigned)g.H) && ((unsigned)x < (unsigned)g.W);
          pidx[r] = ok ? piy[r] * g.W + x : HW;
        }
        bf16x8 aop[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          aop[pl] = lds_tr16x2(img + pl * IPS + pidx[0] * PB + mq * 8, img + pl * IPS + pidx[1] * PB + mq * 8);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) dw[kx][nt] = mfma16_bf16x6(aop, bop[nt], dw[kx][nt]);
      }
    }
    __syncthreads();                                        // the next sample overwrites the planes
  }
  XT_TL(3);
  // ---- this workgroup's slab: dW rows (ky = w, kx, c = 4 g4 + i), columns nt * 16 + c; then the bias gradient
  float* slab = slabs + (size_t)bid * ((size_t)(g.K + 1) * 32);
#pragma unroll
  for (int kx = 0; kx < 4; ++kx)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        slab[(size_t)((w * 4 + kx) * 16 + 4 * g4 + i) * 32 + nt * 16 + c] = dw[kx][nt][i];
  float* red = smem;                                        // [32 thread groups][32 columns] (planes are dead: last barrier)
#pragma unroll
  for (int q = 0; q < 4; ++q) red[(t >> 3) * 32 + (t & 7) * 4 + q] = bsum[q];
  __syncthreads();
  if (t < 32) {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) sum += red[r * 32 + t];
    slab[(size_t)g.K * 32 + t] = sum;
  }
  XT_TL(4);
  XT_TL_DRAIN(5);
}

}  // namespace xt
