// Register-direct implicit GEMM for gfx950 (MI355X, CDNA4): Conv2D/Dense forward, input-gradient and
// weight-gradient on v_mfma_f32_32x32x2_f32 with NO LDS staging and NO barrier in the reduction loop.
//
// Why: the per-block timelines (tools/timeline.py, profiles/r01_timeline.txt) showed that the LDS-tiled kernels
// of xt_igemm.hip keep the matrix pipe 35-60 % busy inside their reduction loop and not at all during their
// 1.5-3.7 us prologues / 2.5-5 us epilogues, because every co-resident block runs the same phase at the same
// time and the waves of a block are chained to each other by a barrier per 32-deep step.  The learner's GEMMs
// are small (0.5-1.7 GFLOP per launch): what they need is many independent waves, not big shared tiles.
//
// How: the reduction order inside a 32-deep step is free as long as both operands agree, so it is chosen such
// that every lane's share of an MFMA operand is CONTIGUOUS in HBM:
//   lane (il = lane & 31, kl = lane >> 5) feeds MFMA kk (0..15) of a step with reduction index 16*kl + kk.
//   fwd   A[m, k]   : 16 consecutive k of one im2col row = 64 contiguous bytes (NHWC, C % 16 == 0) -> 4 x dwordx4
//         B[k, n]   : W[(16 kl + kk) * N + n]: 32 consecutive n per instruction (two full 128-B lines)
//   dgrad A = dY    : 16 consecutive output channels of the tap's output pixel            -> 4 x dwordx4
//         B = W^T   : W[tap][c][16 kl + kk]: contiguous in HWIO                            -> 4 x dwordx4
//   wgrad A = X^T   : X[row(m0 + 16 kl + kk)][k0 + il]: 32 consecutive channels per instruction
//         B = dY    : dY[(m0 + 16 kl + kk) * N + n0 + il]
// Operands go global -> VGPR -> MFMA.  A wave owns a (32 TI) x (32 TJ) output tile over a slice of the
// reduction range with a two-stage register pipeline; the NW waves of a block own different slices of the
// same tile and are combined once, through LDS, in a fixed order (deterministic), and then ALL of them take
// part in the epilogue.  Waves never wait for each other inside the loop, so the 3-5 resident waves of a SIMD
// drift apart and cover each other's load latency.  Blocks are mapped to tiles XCD-contiguously (the 8 XCDs
// have private L2s; neighbouring tiles share im2col lines).
//
// Same arithmetic as xt_igemm.hip (fp32 products, fp32 accumulate, fixed summation order per launch
// configuration).  Shapes outside the envelope (uint8 input, C % 16, K % 32, N % 32) stay on xt_igemm.hip /
// xt_conv1.hip; the launchers return -1 for those.
#include <stdlib.h>
#include "xt_common.h"
#include "xt_igemm.h"
#include "xt_direct_dev.h"

namespace xt {

// Fixed-order combine of the NW per-wave partial tiles (red: [NW][R][64] floats), emitted as 16-byte row pieces:
// red[q][r][kl*32 + il] holds element (row(r, kl), column il) of wave q's partial tile, so four consecutive columns of one row are contiguous in LDS.  Slot e of the R*16 float4
// slots: columns 4*(e & 7).., kl = (e >> 3) & 1, r = e >> 4 -> eight consecutive lanes cover one 128-byte row of a
// 32-column tile (the dword form stored two rows per instruction).  Summation order q = 0..NW-1.
// emit4(r, kl, c4, v): r = tile * 16 + reg.
template <int R, typename F>
__device__ __forceinline__ void combine_and_emit4(float* red, const float (&flat)[R], int w, int NW, int lane, F emit4) {
#pragma unroll
  for (int r = 0; r < R; ++r) red[(w * R + r) * 64 + lane] = flat[r];
  __syncthreads();
  for (int e = w * 64 + lane; e < R * 16; e += NW * 64) {
    const int c4 = (e & 7) * 4, kl = (e >> 3) & 1, r = e >> 4;
    float4 v = *reinterpret_cast<const float4*>(&red[r * 64 + kl * 32 + c4]);
    for (int q = 1; q < NW; ++q) {
      const float4 u = *reinterpret_cast<const float4*>(&red[(q * R + r) * 64 + kl * 32 + c4]);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    emit4(r, kl, c4, v);
  }
}

// ------------------------------------------------------------------ forward
struct DFwdArgs {
  Geom g;
  const float* in;
  const float* w;
  const float* bias;
  float* y;          // ksplit == 1: output; else partial [ksplit][M][N]
  int mt, nt;        // tile grid
  int ksplit;        // cross-block split of the reduction
  int steps_blk;     // 32-deep steps per block
  int nsteps;        // K / 32
};

template <int TI, int TJ, bool PADDED, int MAXT>
__global__ __launch_bounds__(MAXT) void direct_fwd_kernel(const DFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const Geom& g = p.g;
  // the wave index is made explicitly wave-uniform (SGPR): the step loop then runs on the scalar unit
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;
  const int il = lane & 31, kl = lane >> 5;
  XT_TL(0);
  XT_TL_ROLE(70);
  uint32_t lin = xcd_chunk(blockIdx.x, gridDim.x);
  const int z = (int)(lin % (uint32_t)p.ksplit);
  lin /= (uint32_t)p.ksplit;
  const int tn = (int)(lin % (uint32_t)p.nt), tm = (int)(lin / (uint32_t)p.nt);
  const int m0 = tm * (32 * TI), n0 = tn * (32 * TJ);
  const int sb0 = z * p.steps_blk, sb1 = min(p.nsteps, sb0 + p.steps_blk);
  const int per = (sb1 - sb0 + NW - 1) / NW;
  const int s0 = sb0 + w * per, s1 = min(sb1, s0 + per);

  int rowbase[TI], iy0[TI], ix0[TI];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
    const int m = min(m0 + 32 * ti + il, g.M - 1);
    const uint32_t b = fdiv((uint32_t)m, g.d_ohow);
    const uint32_t rem = (uint32_t)m - b * (uint32_t)g.OHOW;
    const uint32_t oy = fdiv(rem, g.d_ow), ox = rem - oy * (uint32_t)g.OW;
    iy0[ti] = (int)oy * g.S - g.PT;
    ix0[ti] = (int)ox * g.S - g.PL;
    rowbase[ti] = (int)b * g.HWC + (iy0[ti] * g.W + ix0[ti]) * g.C;
  }
  const uint32_t wvoff = (uint32_t)(16 * kl * g.N + n0 + il) * 4u;     // lane part of the weight address (fixed)
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, (uint32_t)g.B * (uint32_t)g.HWC * 4u);
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (uint32_t)g.K * (uint32_t)g.N * 4u);

  struct Stage { float4 a[TI][4]; float b[TJ][16]; };
  auto load = [&](Stage& R, int s, bool live) {     // !live: every offset out of range -> zeros, no memory traffic
    const uint32_t k = (uint32_t)(32 * s + 16 * kl);
    const uint32_t ky = fdiv(k, g.d_kwc);
    const uint32_t r = k - ky * (uint32_t)g.KWC;
    const uint32_t kx = fdiv(r, g.d_c);
    const int koff = ((int)ky * g.W + (int)kx) * g.C + (int)(r - kx * (uint32_t)g.C);
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      bool ok = live;
      if (PADDED)
        ok = ok && ((unsigned)(iy0[ti] + (int)ky) < (unsigned)g.H) && ((unsigned)(ix0[ti] + (int)kx) < (unsigned)g.W);
      const uint32_t off = ok ? (uint32_t)(rowbase[ti] + koff) * 4u : kOob;
#pragma unroll
      for (int q = 0; q < 4; ++q) R.a[ti][q] = buf_load4(rs_in, off + 16u * q, 0u);
    }
    // dead stage: out of range through the LANE offset (the range check covers voffset + immediate, not soffset)
    const uint32_t ws = live ? (uint32_t)(32 * s) * (uint32_t)g.N * 4u : 0u;          // uniform
    const uint32_t vo = live ? wvoff : kOob;
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) R.b[tj][kk] = buf_load1(rs_w, vo + 128u * tj, ws + (uint32_t)(kk * g.N) * 4u);
  };
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
  auto compute = [&](const Stage& R) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[TI];
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        const float4 v = R.a[ti][kk >> 2];
        a[ti] = (kk & 3) == 0 ? v.x : (kk & 3) == 1 ? v.y : (kk & 3) == 2 ? v.z : v.w;
      }
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti], R.b[tj][kk], acc[ti][tj], 0, 0, 0);
    }
  };

  // Two register stages.  Every load of the loop is UNCONDITIONAL so that the in-order vmcnt bookkeeping is
  // static (a conditional prefetch makes the compiler wait for everything in flight); prefetches past the end
  // of the wave's slice are issued out of range (no memory access).
  Stage R0, R1;
  load(R0, s0, s0 < s1);
  load(R1, s0 + 1, s0 + 1 < s1);
  XT_TL(1);
  for (int s = s0; s < s1; s += 2) {
    compute(R0);
    load(R0, s + 2, s + 2 < s1);
    compute(R1);                      // an out-of-range stage holds zeros
    load(R1, s + 3, s + 3 < s1);
  }
  XT_TL(3);

  constexpr int R = TI * TJ * 16;
  float flat[R];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) flat[(ti * TJ + tj) * 16 + r] = acc[ti][tj][r];
  const bool final_out = (p.ksplit == 1);
  float* out = final_out ? p.y : p.y + (size_t)z * (size_t)g.M * g.N;
  combine_and_emit4<R>(red, flat, w, NW, lane, [&](int r, int rkl, int c4, float4 v) {
    const int t = r >> 4, rr = r & 15;
    const int ti = t / TJ, tj = t - ti * TJ;
    const int m = m0 + 32 * ti + (rr & 3) + 8 * (rr >> 2) + 4 * rkl;
    const int n = n0 + 32 * tj + c4;
    if (m < g.M) {
      if (final_out) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
        v.x = act_apply(v.x + bv.x, g.act); v.y = act_apply(v.y + bv.y, g.act);
        v.z = act_apply(v.z + bv.z, g.act); v.w = act_apply(v.w + bv.w, g.act);
      }
      store4_wt(out, (size_t)m * g.N + n, v);
    }
  });
  XT_TL(4);
  XT_TL_DRAIN(5);
}

// ------------------------------------------------------------------ input gradient (body: xt_direct_dev.h)
template <int TI, int TJ, int NW>
__global__ __launch_bounds__(64 * NW) void direct_dgrad_kernel(const DDgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  direct_dgrad_body<TI, TJ, NW>(p, blockIdx.x, gridDim.x, red);
}

XT_TL_SETTER(direct)

// ------------------------------------------------------------------ host side
static bool use_direct() { return tuning().direct != 0; }
// Resident-wave target of one launch.  Measured on MI355X (profiles/r01_direct_sweep.txt, r01_timeline.txt): the
// register-direct kernels are bounded by the L2->L1 fill rate (~25 B/clk/CU for row-gathered dwordx4), not by
// latency, so FEW LONG waves (1-2 per SIMD, each streaming its whole reduction slice) beat many short ones.
static int want_waves() { return tuning().direct_waves; }
// direct_all routes every shape inside the envelope to the direct kernels (experiments); by default only
// the shapes that measured faster than the LDS-tiled kernels are: single-column tiles (N or C not a multiple of
// 64, where the LDS tile cannot share the A operand between column tiles anyway) with a long reduction.
static bool direct_all() { return tuning().direct_all != 0; }

static bool direct_envelope(const Geom& g, const xt_input_xform* xf) {
  if (xf && (xf->is_u8 || fabsf(g.xs - 1.f) > 0.f || fabsf(g.xb) > 0.f)) return false;
  return g.C % 16 == 0 && g.K % 32 == 0 && g.N % 32 == 0;
}

// returns -1 when the shape is outside the envelope (caller falls back to the LDS-tiled kernel)
int launch_fwd_direct(const xt_conv_geom* cg, const xt_input_xform* xf, int B, const void* in, const int32_t* idx,
                      const float* w, const float* bias, float* y, float* partial, int ksplit_max, hipStream_t st,
                      int* ksplit_out) {
  if (!use_direct() || idx || !tuning().direct_fwd) return -1;
  DFwdArgs a;
  if (make_geom(cg, xf, B, &a.g)) return -1;
  const Geom& g = a.g;
  if (!direct_envelope(g, xf)) return -1;
  const int TJ = (g.N % 64 == 0) ? 2 : 1;
  if (!direct_all() && (TJ != 1 || g.K < 256)) return -1;
  a.in = static_cast<const float*>(in); a.w = w; a.bias = bias;
  a.mt = (g.M + 31) / 32; a.nt = g.N / (32 * TJ);
  a.nsteps = g.K / 32;
  const int tiles = a.mt * a.nt;
  // reduction slices: enough waves to fill the chip, each with >= 2 steps
  int slices = (want_waves() + tiles - 1) / tiles;
  slices = max(1, min(slices, a.nsteps / 2));
  int nw = min(slices, tuning().direct_max_waves);
  int ks = (slices + nw - 1) / nw;
  if (!partial || ksplit_max < 1) ksplit_max = 1;
  ks = max(1, min(ks, ksplit_max));
  a.steps_blk = (a.nsteps + ks - 1) / ks;
  ks = (a.nsteps + a.steps_blk - 1) / a.steps_blk;
  nw = max(1, min(nw, a.steps_blk));
  a.ksplit = ks;
  a.y = ks == 1 ? y : partial;
  const bool pad = is_padded(g);
  const dim3 grid(tiles * ks), blk(64 * nw);
  const size_t sm = (size_t)nw * TJ * 16 * 64 * sizeof(float);
#define XT_DF(TJV)                                                                                          \
  do {                                                                                                      \
    if (pad) hipLaunchKernelGGL((direct_fwd_kernel<1, TJV, true, 512>), grid, blk, sm, st, a);              \
    else hipLaunchKernelGGL((direct_fwd_kernel<1, TJV, false, 512>), grid, blk, sm, st, a);                 \
  } while (0)
  if (TJ == 2) XT_DF(2); else XT_DF(1);
#undef XT_DF
  XT_LAUNCH_CHECK();
  *ksplit_out = ks;
  return 0;
}

// Plan for the fused per-layer backward launch (256-thread blocks -> NW = 4, one 32x32 tile per block): only the
// shapes where the direct kernel measured faster (single-column tiles, long reduction, not too many tiles).
bool plan_dgrad_direct_fused(const Geom& g, DDgradArgs* a, int* nblocks) {
  if (!use_direct() || !tuning().direct_dgrad || g.N % 32 != 0 || g.C % 32 != 0) return false;
  if (g.C % 64 == 0) return false;                       // TJ = 2 shapes stay on the LDS-tiled kernel
  const int hc = (g.H + g.S - 1) / g.S, wc = (g.W + g.S - 1) / g.S;
  const int mc = g.B * hc * wc;
  const int nclass = g.S * g.S;
  const int jmax = ((g.KH + g.S - 1) / g.S) * ((g.KW + g.S - 1) / g.S);
  const int nsteps = jmax * (g.N / 32);
  const int tiles = ((mc + 31) / 32) * (g.C / 32) * nclass;
  if (!direct_all() && (nsteps < 8 || tiles >= tuning().direct_tile64_tiles)) return false;
  a->g = g;
  a->mt = (mc + 31) / 32;
  a->ct = g.C / 32;
  *nblocks = tiles;
  return true;
}

int launch_dgrad_direct(const xt_conv_geom* cg, int B, const float* dy, const float* w, const float* x, int act_prev,
                        float* dx, hipStream_t st) {
  if (!use_direct()) return -1;
  DDgradArgs a;
  a.deep = 0;
  if (make_geom(cg, nullptr, B, &a.g)) return -1;
  const Geom& g = a.g;
  if (g.N % 32 != 0 || g.C % 32 != 0) return -1;
  a.dy = dy; a.w = w; a.x = x; a.dx = dx; a.act_prev = act_prev;
  const int TJ = (g.C % 64 == 0) ? 2 : 1;
  const int hc = (g.H + g.S - 1) / g.S, wc = (g.W + g.S - 1) / g.S;   // upper bound on class extent
  const int mc = B * hc * wc;
  const int nclass = g.S * g.S;
  const int jmax = ((g.KH + g.S - 1) / g.S) * ((g.KW + g.S - 1) / g.S);
  const int nsteps = jmax * (g.N / 32);
  const int tiles1 = ((mc + 31) / 32) * (g.C / (32 * TJ)) * nclass;
  const int TI = (tiles1 >= tuning().direct_tile64_tiles) ? 2 : 1;
  if (!direct_all() && (TJ != 1 || TI != 1 || nsteps < 8)) return -1;
  a.mt = (mc + 32 * TI - 1) / (32 * TI);
  a.ct = g.C / (32 * TJ);
  const int tiles = a.mt * a.ct * nclass;
  int nw = want_waves() / max(1, tiles);
  nw = min(nw, nsteps / 2);
  nw = nw >= 4 ? 4 : nw >= 2 ? 2 : 1;
  const dim3 grid(tiles), blk(64 * nw);
  const size_t sm = (size_t)nw * TI * TJ * 16 * 64 * sizeof(float);
#define XT_DD(TIV, TJV)                                                                                     \
  do {                                                                                                      \
    if (nw == 4) hipLaunchKernelGGL((direct_dgrad_kernel<TIV, TJV, 4>), grid, blk, sm, st, a);              \
    else if (nw == 2) hipLaunchKernelGGL((direct_dgrad_kernel<TIV, TJV, 2>), grid, blk, sm, st, a);         \
    else hipLaunchKernelGGL((direct_dgrad_kernel<TIV, TJV, 1>), grid, blk, sm, st, a);                      \
  } while (0)
  if (TI == 2 && TJ == 2) XT_DD(2, 2);
  else if (TI == 2) XT_DD(2, 1);
  else if (TJ == 2) XT_DD(1, 2);
  else XT_DD(1, 1);
#undef XT_DD
  XT_LAUNCH_CHECK();
  return 0;
}

}  // namespace xt
