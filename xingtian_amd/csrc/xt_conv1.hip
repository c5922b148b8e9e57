// First-layer (uint8 frame-stack) convolution kernels on the bf16 matrix cores with an EXACT 3-way split.
//
// An 8-bit pixel is exactly representable in bf16 (8 significant bits).  An fp32 weight w is split into three
// bf16 planes w = w1 + w2 + w3 (truncation splits, 3 x 8 = 24 significant bits = all of fp32), so
//     sum_k x_k * w_k  =  sum_k x_k*w1_k + x_k*w2_k + x_k*w3_k
// with every product exact in fp32 (8 + 8 significant bits) and fp32 accumulation inside the MFMA: the result
// has fp32-class accuracy (only the summation order differs from a scalar fp32 loop) while running on
// v_mfma_f32_32x32x16_bf16 (3 x 32 cycles per 32x32x16) instead of v_mfma_f32_32x32x2_f32 (8 x 64 cycles) =
// 16/3 x the fp32 matrix rate.  The x/255 (or (x-mean)/std) transform is applied to the accumulator.
//
// Geometry handled: uint8 NHWC input with C = 4, KW = 8 (one kernel row = 32 contiguous bytes), N = 32 output
// channels, no padding (TF VALID) -- PpoCnn's 8x8/4 first layer on 84x84x4 frame stacks
// (xt/model/model_utils.py:126-131).  One workgroup stages ONE frame stack (28 KB) into LDS with coalesced
// 16-byte loads (the minibatch row gather idx[b] is fused here) and produces all of its OH*OW x 32 outputs.
#include "xt_common.h"
#include "xt_conv1_dev.h"

namespace xt {

struct C1FwdArgs {
  const uint8_t* in;
  const int32_t* idx;
  const float* w;      // [KH*32][32]
  const float* bias;   // [32]
  float* y;            // [B*OH*OW][32]
  int B, H, W, OH, OW, S, KH, act;
  float xs, xb;        // y = act(acc*xs + bias)   (xb must be 0 for this kernel)
  int img_cap;         // flattened form: LDS bytes of the bf16 image of the packed input rows (multiple of 16)
  uint32_t* mask;      // flattened form, optional: one word per output position, bit n = (output channel n > 0)
};

constexpr int kC1TileSlots = 16;   // 32-pixel tile slots per workgroup -> OH*OW <= 512

// NW waves per workgroup, 16/NW tile slots per wave.  Two workgroups fit a CU (LDS) for either NW; with NW = 8 a
// SIMD holds two waves per workgroup, so the LDS reads and byte->bf16 conversions of one wave are issued under the
// MFMAs of the other (a lone wave issues in order: its step took 1065 cycles for 384 cycles of MFMA).
template <int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void conv_u8c4k8_fwd_bf16x3_kernel(const C1FwdArgs p) {
  constexpr int NT = 64 * NW, kC1MaxTiles = kC1TileSlots / NW, WQ = 1024 / NT;
  extern __shared__ __attribute__((aligned(16))) uint8_t limg[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int b = blockIdx.x;
  const int HWC = p.H * p.W * 4, Wrow = p.W * 4, OHOW = p.OH * p.OW;
  const int nsteps = 2 * p.KH;              // 16 reduction elements per step = half a kernel row
  uint4* wpl = reinterpret_cast<uint4*>(limg + HWC);     // [nsteps][3 planes][64 lanes] x 16 B, MFMA operand order
  XT_TL(0);
  XT_TL_ROLE(40);
  // ---- issue every global load of the block up front: this thread's share of the weights (fp32, split below)
  // and of the frame stack (minibatch gather fused) -> one memory latency for the whole prologue
  float wv[WQ][8];
  const int nslots = nsteps * 64;
#pragma unroll
  for (int q = 0; q < WQ; ++q) {
    const int slot = t + NT * q;
    const int sc = slot < nslots ? slot : 0;
    const float* wl = p.w + (size_t)((sc >> 6) * 16 + 8 * ((sc & 63) >> 5)) * 32 + (sc & 31);
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[q][j] = wl[j * 32];
  }
  {
    const size_t s = p.idx ? (size_t)p.idx[b] : (size_t)b;
    const uint4* src = reinterpret_cast<const uint4*>(p.in + s * (size_t)HWC);
    stage_image<NT>(src, reinterpret_cast<uint4*>(limg), HWC >> 4, t);
  }
  XT_TL(1);
  // exact 3-way bf16 split of the weights, written once per block in operand order
#pragma unroll
  for (int q = 0; q < WQ; ++q) {
    const int slot = t + NT * q;
    if (slot < nslots) {
      BF8 b1, b2, b3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float w0 = wv[q][2 * e], w1 = wv[q][2 * e + 1];
        const float r0 = w0 - trunc_bf16(w0), r1 = w1 - trunc_bf16(w1);
        const float q0 = r0 - trunc_bf16(r0), q1 = r1 - trunc_bf16(r1);
        b1.u[e] = pack_hi16(w0, w1);
        b2.u[e] = pack_hi16(r0, r1);
        b3.u[e] = pack_hi16(q0, q1);
      }
      const int s = slot >> 6, ln = slot & 63;
      wpl[(s * 3 + 0) * 64 + ln] = make_uint4(b1.u[0], b1.u[1], b1.u[2], b1.u[3]);
      wpl[(s * 3 + 1) * 64 + ln] = make_uint4(b2.u[0], b2.u[1], b2.u[2], b2.u[3]);
      wpl[(s * 3 + 2) * 64 + ln] = make_uint4(b3.u[0], b3.u[1], b3.u[2], b3.u[3]);
    }
  }
  const int ntiles = (OHOW + 31) >> 5;
  const int il = lane & 31, h = lane >> 5;
  // per-tile byte offset of this lane's pixel (top-left of its receptive field)
  int poff[kC1MaxTiles];
#pragma unroll
  for (int ti = 0; ti < kC1MaxTiles; ++ti) {
    const int pix = (wave + NW * ti) * 32 + il;
    const int pp = pix < OHOW ? pix : 0;
    const int oy = pp / p.OW, ox = pp - oy * p.OW;
    poff[ti] = (p.S * oy * p.W + p.S * ox) * 4 + 8 * h;
  }
  f32x16 acc[kC1MaxTiles];
#pragma unroll
  for (int ti = 0; ti < kC1MaxTiles; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;
  const float bias = p.bias[il];
  __syncthreads();
  XT_TL(2);

  // software-pipelined main loop: the LDS reads of step s+1 (3 weight planes + one 8-byte pixel group per
  // tile) are issued before the MFMAs of step s; MFMAs are issued plane-major so that consecutive
  // instructions hit different accumulators (no dependent-accumulator stall).
  uint4 wq[3];
  uint2 aq[kC1MaxTiles];
  auto lds_fetch = [&](int s) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) wq[pl] = wpl[(s * 3 + pl) * 64 + lane];
    const int koff = (s >> 1) * Wrow + (s & 1) * 16;
#pragma unroll
    for (int ti = 0; ti < kC1MaxTiles; ++ti)
      aq[ti] = *reinterpret_cast<const uint2*>(limg + poff[ti] + koff);   // tiles beyond ntiles read pixel 0: harmless
  };
  lds_fetch(0);
  for (int s = 0; s < nsteps; ++s) {
    BF8 bp[3];
    bf16x8 av[kC1MaxTiles];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) { bp[pl].u[0] = wq[pl].x; bp[pl].u[1] = wq[pl].y; bp[pl].u[2] = wq[pl].z; bp[pl].u[3] = wq[pl].w; }
#pragma unroll
    for (int ti = 0; ti < kC1MaxTiles; ++ti) av[ti] = bytes_to_bf16x8(aq[ti].x, aq[ti].y);
    if (s + 1 < nsteps) lds_fetch(s + 1);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int ti = 0; ti < kC1MaxTiles; ++ti)      // unconditional: a tile slot beyond ntiles recomputes pixel 0 and is
        acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ti], bp[pl].v, acc[ti], 0, 0, 0);   // dropped in the epilogue;
  }                                                  // a (wave-uniform) branch per MFMA breaks their back-to-back issue

  // ---- epilogue: input scale on the accumulator, bias, activation.  The 32x32 fp32 tile of a wave is ONE
  // contiguous 4 KB block of the NHWC output (N = 32): transpose it through LDS (the weight-plane region is
  // dead after a barrier) and write it with 16-byte-per-lane stores (4 per tile instead of 16 scattered
  // dword stores: the store phase was 8.5 of the kernel's 23 us).
  __syncthreads();
  XT_TL(3);
  float* tbuf = reinterpret_cast<float*>(limg + HWC) + wave * (32 * 36);      // [32 rows][36] padded, per wave
#pragma unroll
  for (int ti = 0; ti < kC1MaxTiles; ++ti) {
    if (wave + NW * ti < ntiles) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        tbuf[row * 36 + il] = act_apply(fmaf(acc[ti][r], p.xs, bias), p.act);
      }
      // wave-private region: program order of one wave suffices between its own ds_write and ds_read
      const int pix0 = (wave + NW * ti) * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = q * 8 + (lane >> 3), c4 = (lane & 7) * 4;
        const float4 v = *reinterpret_cast<const float4*>(&tbuf[row * 36 + c4]);
        if (pix0 + row < OHOW)
          *reinterpret_cast<float4*>(&p.y[((size_t)b * OHOW + pix0 + row) * 32 + c4]) = v;
      }
    }
  }
  XT_TL(4);
  XT_TL_DRAIN(5);
}

// Stage the input rows that the output positions [p0, p1) of the flattened [B*OH*OW] range need (the range touches at
// most NIMG frame stacks, first one = p0 / OHOW; minibatch row gather fused), the per-stack row groups packed back to
// back.  All loads of a pass are issued back to back (offsets relative to `in`, not per-stack pointers: a select
// between pointers degrades to flat loads; unconditional LDS writes: a guarded one makes hipcc sink each load into its
// branch); shift[i] + o is the LDS byte of byte o of stack i.  BF16: every byte is converted to bf16 on the
// way (exact), i.e. the image occupies 2 bytes per element and LDS byte = 2 * (shift + o).
template <int NIMG, int NT, int U, bool BF16 = false>
__device__ __forceinline__ void stage_position_rows_packed(const uint8_t* __restrict__ in, const int32_t* __restrict__ idx,
                                                           int B, int HWC, int Wrow, int OHOW, int OW, int S, int KH,
                                                           int p0, int p1, uint8_t* limg, int t, int (&shift)[NIMG]) {
  const int s0 = p0 / OHOW, slast = (p1 - 1) / OHOW;
  int un[NIMG], dbase[NIMG], srow[NIMG];
  long long goff[NIMG];
  int ntot = 0;
#pragma unroll
  for (int i = 0; i < NIMG; ++i) {
    const int sc = min(s0 + i, B - 1);
    srow[i] = idx ? idx[sc] : sc;
  }
#pragma unroll
  for (int i = 0; i < NIMG; ++i) {
    const int sidx = s0 + i;
    un[i] = 0; dbase[i] = 0; goff[i] = 0; shift[i] = 0;
    if (sidx <= slast) {
      const int lo = max(p0, sidx * OHOW) - sidx * OHOW, hi = min(p1, (sidx + 1) * OHOW) - 1 - sidx * OHOW;
      const int blo = S * (lo / OW) * Wrow, bhi = (S * (hi / OW) + KH) * Wrow;
      const int ul = blo >> 4;
      un[i] = ((bhi + 15) >> 4) - ul;
      goff[i] = (long long)srow[i] * (long long)HWC + (long long)ul * 16;
      dbase[i] = ntot * 16;
      shift[i] = (ntot - ul) * 16;          // LDS byte of the stack's byte o: shift + o
    }
    ntot += un[i];
  }
  for (int base = 0; base < ntot; base += NT * U) {
    uint4 v[U];
    int dsto[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      int u = base + t + NT * q;
      const bool ok = u < ntot;
      u = ok ? u : 0;
      long long go = goff[0];
      int db = dbase[0];
#pragma unroll
      for (int j = 0; j + 1 < NIMG; ++j) {
        int cum = 0;
#pragma unroll
        for (int k = 0; k <= j; ++k) cum += un[k];
        if (u >= cum) { go = goff[j + 1] - (long long)cum * 16; db = dbase[j + 1] - cum * 16; }
      }
      v[q] = *reinterpret_cast<const uint4*>(in + go + (long long)u * 16);
      dsto[q] = db + u * 16;
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      if constexpr (BF16) {
        BF8 lo, hi;
        lo.v = bytes_to_bf16x8(v[q].x, v[q].y);
        hi.v = bytes_to_bf16x8(v[q].z, v[q].w);
        *reinterpret_cast<uint4*>(limg + 2 * dsto[q]) = make_uint4(lo.u[0], lo.u[1], lo.u[2], lo.u[3]);
        *reinterpret_cast<uint4*>(limg + 2 * dsto[q] + 16) = make_uint4(hi.u[0], hi.u[1], hi.u[2], hi.u[3]);
      } else {
        *reinterpret_cast<uint4*>(limg + dsto[q]) = v[q];
      }
    }
  }
}

// bytes of packed input rows the largest 512-position range of the launch stages (host side)
static int c1_packed_cap(int B, int OH, int OW, int S, int KH, int Wrow, int PB) {
  const int OHOW = OH * OW, total = B * OHOW;
  int cap = 0;
  for (int p0 = 0; p0 < total; p0 += PB) {
    const int p1 = total < p0 + PB ? total : p0 + PB;
    int units = 0;
    for (int sidx = p0 / OHOW; sidx <= (p1 - 1) / OHOW; ++sidx) {
      const int lo = (p0 > sidx * OHOW ? p0 : sidx * OHOW) - sidx * OHOW;
      const int hi = (p1 < (sidx + 1) * OHOW ? p1 : (sidx + 1) * OHOW) - 1 - sidx * OHOW;
      const int blo = S * (lo / OW) * Wrow, bhi = (S * (hi / OW) + KH) * Wrow;
      units += ((bhi + 15) >> 4) - (blo >> 4);
    }
    if (units * 16 > cap) cap = units * 16;
  }
  return cap;
}

// ---- position-flattened forward: one workgroup = PB consecutive output positions of the flattened [B*OH*OW]
// range instead of one frame stack.  With one frame stack per workgroup, B = 320 on 256 CUs leaves 64 CUs with two
// co-resident workgroups (13.9 us) while the other 192 finish their single one in 9.9 us (profiles/r01_timeline*);
// 512 positions per workgroup are exactly 16 tiles = 8 waves x 2 slots (no idle tile slot: 13 tiles of a frame stack
// occupied 16) and 250 equal workgroups, one per CU.  A position range touches at most NIMG frame stacks; only the
// input rows its output rows need are staged (same byte offsets inside a per-stack LDS slot).
template <int SLOTS>
__global__ __launch_bounds__(512, 2) void conv_u8c4k8_fwd_flat_kernel(const C1FwdArgs p) {
  constexpr int NW = 8, NT = 512, PB = 32 * NW * SLOTS, NIMG = SLOTS == 2 ? 3 : 2, WQ = 1024 / NT, U = SLOTS == 2 ? 6 : 4;
  extern __shared__ __attribute__((aligned(16))) uint8_t limg[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int HWC = p.H * p.W * 4, Wrow = p.W * 4, OHOW = p.OH * p.OW;
  const int total = p.B * OHOW;
  const int p0 = blockIdx.x * PB, p1 = min(total, p0 + PB);
  const int nsteps = 2 * p.KH;
  uint4* wpl = reinterpret_cast<uint4*>(limg + p.img_cap);     // behind the bf16 image of the packed input rows
  XT_TL(0);
  XT_TL_ROLE(40);
  float wv[WQ][8];
  const int nslots = nsteps * 64;
#pragma unroll
  for (int q = 0; q < WQ; ++q) {
    const int slot = t + NT * q;
    const int sc = slot < nslots ? slot : 0;
    const float* wl = p.w + (size_t)((sc >> 6) * 16 + 8 * ((sc & 63) >> 5)) * 32 + (sc & 31);
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[q][j] = wl[j * 32];
  }
  // The input rows are converted to bf16 ONCE while they are staged (2 bytes per element, packed rows: <= 88 KB):
  // the loop's A operand is then one ds_read_b128 per tile.  Converting at operand-read time repeated every byte's
  // conversion four times (8x8 windows at stride 4) and cost 24 VALU instructions per step next to 6 MFMAs.
  int shift[NIMG];
  stage_position_rows_packed<NIMG, NT, U, true>(p.in, p.idx, p.B, HWC, Wrow, OHOW, p.OW, p.S, p.KH, p0, p1, limg, t, shift);
  const int s0 = p0 / OHOW;
  XT_TL(1);
#pragma unroll
  for (int q = 0; q < WQ; ++q) {
    const int slot = t + NT * q;
    if (slot < nslots) {
      BF8 b1, b2, b3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float w0 = wv[q][2 * e], w1 = wv[q][2 * e + 1];
        const float r0 = w0 - trunc_bf16(w0), r1 = w1 - trunc_bf16(w1);
        const float q0 = r0 - trunc_bf16(r0), q1 = r1 - trunc_bf16(r1);
        b1.u[e] = pack_hi16(w0, w1);
        b2.u[e] = pack_hi16(r0, r1);
        b3.u[e] = pack_hi16(q0, q1);
      }
      const int sidx = slot >> 6, ln = slot & 63;
      wpl[(sidx * 3 + 0) * 64 + ln] = make_uint4(b1.u[0], b1.u[1], b1.u[2], b1.u[3]);
      wpl[(sidx * 3 + 1) * 64 + ln] = make_uint4(b2.u[0], b2.u[1], b2.u[2], b2.u[3]);
      wpl[(sidx * 3 + 2) * 64 + ln] = make_uint4(b3.u[0], b3.u[1], b3.u[2], b3.u[3]);
    }
  }
  const int il = lane & 31, h = lane >> 5;
  int poff[SLOTS];                           // LDS byte of this lane's 8 bf16 at step 0
#pragma unroll
  for (int ti = 0; ti < SLOTS; ++ti) {
    const int pp = min(p0 + (wave + NW * ti) * 32 + il, p1 - 1);      // tail positions recompute the last valid one
    const int sidx = pp / OHOW, rem = pp - sidx * OHOW;
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    const int di = sidx - s0;
    int sh = shift[0];
#pragma unroll
    for (int j = 1; j < NIMG; ++j) sh = di == j ? shift[j] : sh;
    poff[ti] = 2 * (sh + (p.S * oy * p.W + p.S * ox) * 4 + 8 * h);
  }
  f32x16 acc[SLOTS];
#pragma unroll
  for (int ti = 0; ti < SLOTS; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;
  const float bias = p.bias[il];
  __syncthreads();
  XT_TL(2);
  // The step count is a compile-time constant (KH = 8 -> 16 half kernel rows) and the loop is fully unrolled with
  // two explicit operand register sets: as a rolled loop hipcc rotated the pipeline with 16 v_mov per step and
  // waited lgkmcnt(0) for the NEXT step's operands right behind the first MFMA of the current one (ISA), i.e. the
  // LDS latency was exposed once per step.
  constexpr int NS = 16;
  uint4 wq[2][3], aq[2][SLOTS];
  auto lds_fetch = [&](int s, uint4 (&w3)[3], uint4 (&a2)[SLOTS]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) w3[pl] = wpl[(s * 3 + pl) * 64 + lane];
    const int koff = 2 * ((s >> 1) * Wrow + (s & 1) * 16);
#pragma unroll
    for (int ti = 0; ti < SLOTS; ++ti) a2[ti] = *reinterpret_cast<const uint4*>(limg + poff[ti] + koff);
  };
  lds_fetch(0, wq[0], aq[0]);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int cur = s & 1;
    BF8 bp[3], av[SLOTS];
    // the NEXT step's LDS reads go out first and stay there (sched_barrier: left alone, hipcc sinks each read to just
    // in front of its use to save registers and then waits for it behind one MFMA)
    if (s + 1 < NS) lds_fetch(s + 1, wq[cur ^ 1], aq[cur ^ 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) { bp[pl].u[0] = wq[cur][pl].x; bp[pl].u[1] = wq[cur][pl].y; bp[pl].u[2] = wq[cur][pl].z; bp[pl].u[3] = wq[cur][pl].w; }
#pragma unroll
    for (int ti = 0; ti < SLOTS; ++ti) { av[ti].u[0] = aq[cur][ti].x; av[ti].u[1] = aq[cur][ti].y; av[ti].u[2] = aq[cur][ti].z; av[ti].u[3] = aq[cur][ti].w; }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int ti = 0; ti < SLOTS; ++ti)
        acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ti].v, bp[pl].v, acc[ti], 0, 0, 0);
  }
  __syncthreads();
  XT_TL(3);
  float* tbuf = reinterpret_cast<float*>(limg + p.img_cap) + wave * (32 * 36);
#pragma unroll
  for (int ti = 0; ti < SLOTS; ++ti) {
    const int pix0 = p0 + (wave + NW * ti) * 32;
    if (pix0 < p1) {
      // relu (un-switched by hand: with the activation selected per element the loop kept a branch tree and a tanh
      // CALL per value) + the sign mask of the outputs for the consumer's input gradient, which needs relu'(y) only
      // -- one word per position instead of a 128-byte row: a ballot over the wave is the 32 channel bits of two rows
      uint32_t mw = 0u;
      if (p.act == XT_ACT_RELU) {
        // word of row L lands in lane L: v_writelane_b32 moves the (wave-uniform) ballot halves into lanes row0 and
        // row0 + 4 -- two instructions per ballot; selecting with per-lane compares took six
#define XT_C1_RELU_ROW(r)                                                                                        \
        {                                                                                                        \
          constexpr int row0 = ((r) & 3) + 8 * ((r) >> 2);                                                       \
          const float z = fmaf(acc[ti][r], p.xs, bias);                                                          \
          const unsigned long long bal = __ballot(z > 0.f);                                                      \
          tbuf[(row0 + 4 * h) * 36 + il] = z > 0.f ? z : 0.f;                                                    \
          asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(mw) : "s"((uint32_t)bal), "n"(row0));                \
          asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(mw) : "s"((uint32_t)(bal >> 32)), "n"(row0 + 4));    \
        }
        XT_C1_RELU_ROW(0) XT_C1_RELU_ROW(1) XT_C1_RELU_ROW(2) XT_C1_RELU_ROW(3)
        XT_C1_RELU_ROW(4) XT_C1_RELU_ROW(5) XT_C1_RELU_ROW(6) XT_C1_RELU_ROW(7)
        XT_C1_RELU_ROW(8) XT_C1_RELU_ROW(9) XT_C1_RELU_ROW(10) XT_C1_RELU_ROW(11)
        XT_C1_RELU_ROW(12) XT_C1_RELU_ROW(13) XT_C1_RELU_ROW(14) XT_C1_RELU_ROW(15)
#undef XT_C1_RELU_ROW
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tbuf[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + il] = act_apply(fmaf(acc[ti][r], p.xs, bias), p.act);
      }
      if (p.mask && lane < 32 && pix0 + lane < p1) p.mask[pix0 + lane] = mw;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = q * 8 + (lane >> 3), c4 = (lane & 7) * 4;
        const float4 v = *reinterpret_cast<const float4*>(&tbuf[row * 36 + c4]);
        if (pix0 + row < p1) store4_wt(p.y, (size_t)(pix0 + row) * 32 + c4, v);
      }
    }
  }
  XT_TL(4);
  XT_TL_DRAIN(5);
}

static int c1_waves() { return tuning().conv1_waves == 4 ? 4 : 8; }      // 4: the four-wave forms (A/B)

// returns 0 launched, 1 error, -1 geometry not handled by this kernel
int launch_conv1_fwd_bf16x3(const xt_conv_geom* g, const xt_input_xform* xf, int B, const void* in,
                            const int32_t* idx, const float* w, const float* bias, float* y, hipStream_t st,
                            uint32_t* relu_mask, int* mask_written) {
  if (mask_written) *mask_written = 0;
  if (!xf || !xf->is_u8 || g->C != 4 || g->KW != 8 || g->N != 32 || g->PT != 0 || g->PL != 0) return -1;
  if ((g->OH - 1) * g->S + g->KH > g->H || (g->OW - 1) * g->S + g->KW > g->W) return -1;
  const int HWC = g->H * g->W * 4;
  if (HWC % 16 != 0 || HWC > 64 * 1024 || (g->W * 4) % 8 != 0 || (g->S * 4) % 8 != 0) return -1;
  if (g->OH * g->OW > 32 * kC1TileSlots) return -1;
  if (fabsf(xf->mean) >= 1e-4f || g->KH > 8) return -1;     // the mean term would need colsum(W); 4 weight slots/thread
  // store4_wt addresses y with a 32-bit byte offset: leave outputs of 2 GiB or more to the generic path, whose geometry
  // check rejects them loudly (make_geom)
  if ((long long)B * g->OH * g->OW * g->N * 4 >= (1ll << 31)) return -1;
  C1FwdArgs a;
  a.in = static_cast<const uint8_t*>(in); a.idx = idx; a.w = w; a.bias = bias; a.y = y;
  a.B = B; a.H = g->H; a.W = g->W; a.OH = g->OH; a.OW = g->OW; a.S = g->S; a.KH = g->KH; a.act = g->act;
  const float mean = fabsf(xf->mean) >= 1e-4f ? xf->mean : 0.f;
  a.xs = 1.f / xf->std; a.xb = -mean * a.xs; a.img_cap = 0;
  a.mask = (g->act == XT_ACT_RELU) ? relu_mask : nullptr;
  const int nw = c1_waves();
  size_t lds = (size_t)2 * g->KH * 3 * 64 * 16;                    // weight planes, reused by the output transpose
  if (lds < (size_t)nw * 32 * 36 * 4) lds = (size_t)nw * 32 * 36 * 4;
  lds += (size_t)HWC;
  const int flat = tuning().conv1_flat;     // 0: one frame stack per workgroup (A/B)
  if (flat && nw == 8 && g->KH == 8) {
    const int total = B * g->OH * g->OW;
    const bool two = (total + 511) / 512 >= 200;
    const int pb = two ? 512 : 256, nimg = two ? 3 : 2;
    if ((pb - 1) / (g->OH * g->OW) + 2 <= nimg) {          // a range of pb positions touches at most nimg frame stacks
      a.img_cap = 2 * c1_packed_cap(B, g->OH, g->OW, g->S, g->KH, g->W * 4, pb);
      const size_t fl = (size_t)a.img_cap + (size_t)2 * g->KH * 3 * 64 * 16;
      static PerDeviceOnce attr_once;
      attr_once.run([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_u8c4k8_fwd_flat_kernel<2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_u8c4k8_fwd_flat_kernel<1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
      });
      if (fl <= 160 * 1024) {
        if (two) hipLaunchKernelGGL(conv_u8c4k8_fwd_flat_kernel<2>, dim3((total + pb - 1) / pb), dim3(512), fl, st, a);
        else hipLaunchKernelGGL(conv_u8c4k8_fwd_flat_kernel<1>, dim3((total + pb - 1) / pb), dim3(512), fl, st, a);
        XT_LAUNCH_CHECK();
        if (mask_written && a.mask) *mask_written = 1;
        return 0;
      }
    }
  }
  if (nw == 8) hipLaunchKernelGGL(conv_u8c4k8_fwd_bf16x3_kernel<8>, dim3(B), dim3(512), lds, st, a);
  else hipLaunchKernelGGL(conv_u8c4k8_fwd_bf16x3_kernel<4>, dim3(B), dim3(256), lds, st, a);
  XT_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ first-layer weight gradient
// dW[k,n] = sum_pixels x[pixel,k] * dY[pixel,n]  (x uint8 -> exact bf16 A operand, dY split into 3 bf16 planes
// as the B operand; the reduction index of v_mfma_f32_32x32x16_bf16 is 16 pixels per instruction).
// One workgroup = one frame stack (staged into LDS like the forward) -> one partial slab [(K+1)*32]; slabs are
// summed by grads_finish_kernel.  Wave (pg, kh): pixel steps of parity pg, kernel rows [4kh, 4kh+4) (one kernel
// row = one 32-wide k tile); the two pixel-parity halves are combined through LDS.  dY is read straight from
// global/L2 (each lane: 8 pixels x its column n, coalesced 128 B rows) and split in registers once per step for
// all four k tiles; the x operand is gathered with ds_read_u8 (32 consecutive bytes per half-wave: conflict-free).
struct C1WgArgs {
  const uint8_t* in;
  const int32_t* idx;
  const float* dy;     // [B*OH*OW][32] d(pre-activation)
  float* out;          // [B][(K+1)*32] partial slabs
  int B, H, W, OH, OW, S, KH;
  float xs, xb;
  int img_cap;         // flattened form: LDS bytes reserved for the packed input rows (multiple of 16)
};

// NKQ = kernel-row groups per pixel parity: 2 pixel parities x NKQ groups = 2*NKQ waves, 8/NKQ kernel rows (k
// tiles) per wave.  NKQ = 4 puts two waves of a workgroup on every SIMD (see the forward kernel).
template <int NKQ>
__global__ __launch_bounds__(128 * NKQ, NKQ) void conv_u8c4k8_wgrad_bf16x3_kernel(const C1WgArgs p) {
  constexpr int NT = 128 * NKQ, RQ = 8 / NKQ, DQ = 4096 / NT;
  extern __shared__ __attribute__((aligned(16))) uint8_t lsm[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int b = blockIdx.x;
  const int HWC = p.H * p.W * 4, Wrow = p.W * 4, OHOW = p.OH * p.OW;
  const int nsteps = (OHOW + 15) >> 4;
  uint8_t* limg = lsm;
  float* dys = reinterpret_cast<float*>(lsm + HWC);                      // [nsteps*16][32] this sample's dY
  int* pixoff = reinterpret_cast<int*>(lsm + HWC + nsteps * 16 * 32 * 4); // [nsteps*16]
  float* red = dys;                                                      // aliases dys after the main loop
  float* bred = reinterpret_cast<float*>(pixoff);                        // aliases pixoff after the main loop
  XT_TL(0);
  XT_TL_ROLE(50);
  // ---- every global load of the block is issued up front (dY rows of this sample + the frame stack)
  {
    const float4* dsrc = reinterpret_cast<const float4*>(p.dy + (size_t)b * OHOW * 32);
    const int n4 = OHOW * 8, n4pad = nsteps * 16 * 8;
    float4 dv[DQ];
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      const int i = t + NT * q;
      dv[q] = dsrc[i < n4 ? i : 0];
    }
    const size_t s = p.idx ? (size_t)p.idx[b] : (size_t)b;
    const uint4* src = reinterpret_cast<const uint4*>(p.in + s * (size_t)HWC);
    stage_image<NT>(src, reinterpret_cast<uint4*>(limg), HWC >> 4, t);
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      const int i = t + NT * q;
      if (i < n4pad) reinterpret_cast<float4*>(dys)[i] = i < n4 ? dv[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = t; i < nsteps * 16; i += NT) {
      const int pp = i < OHOW ? i : 0;
      const int oy = pp / p.OW, ox = pp - oy * p.OW;
      pixoff[i] = (p.S * oy * p.W + p.S * ox) * 4;
    }
  }
  const int il = lane & 31, h = lane >> 5;
  const int pg = wave / NKQ, kq = wave - pg * NKQ;
  f32x16 acc[RQ];
#pragma unroll
  for (int q = 0; q < RQ; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float bsum = 0.f;
  XT_TL(1);
  __syncthreads();
  XT_TL(2);

  // software-pipelined: LDS reads of this wave's NEXT step (8 dY values, RQ x 8 pixel bytes) are issued before
  // the MFMAs of the current one; the pixel-offset table is read one step further ahead (the byte addresses
  // depend on it).  MFMAs are issued plane-major (consecutive instructions hit different accumulators).
  float dyr[8];
  uint32_t xr[RQ][8];
  int po[8];
  auto read_po = [&](int s) {
#pragma unroll
    for (int e = 0; e < 8; ++e) po[e] = pixoff[s * 16 + 8 * h + e];
  };
  auto read_ops = [&](int s) {     // uses po[] of step s
#pragma unroll
    for (int e = 0; e < 8; ++e) dyr[e] = dys[(s * 16 + 8 * h + e) * 32 + il];
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const int kb = (kq * RQ + q) * Wrow + il;       // kernel row ky = RQ*kq+q, byte kx*4+c = il
#pragma unroll
      for (int e = 0; e < 8; ++e) xr[q][e] = limg[po[e] + kb];
    }
  };
  if (pg < nsteps) { read_po(pg); read_ops(pg); }
  if (pg + 2 < nsteps) read_po(pg + 2);
  for (int s = pg; s < nsteps; s += 2) {
    // exact 3-way bf16 split of the 8 dY values of this lane (pixels 16s+8h+e, column il)
    BF8 bp[3], av[RQ];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float w0 = dyr[2 * q], w1 = dyr[2 * q + 1];
      bsum += w0 + w1;
      const float r0 = w0 - trunc_bf16(w0), r1 = w1 - trunc_bf16(w1);
      const float q0 = r0 - trunc_bf16(r0), q1 = r1 - trunc_bf16(r1);
      bp[0].u[q] = pack_hi16(w0, w1);
      bp[1].u[q] = pack_hi16(r0, r1);
      bp[2].u[q] = pack_hi16(q0, q1);
    }
#pragma unroll
    for (int q = 0; q < RQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) av[q].u[e] = pack_hi16((float)xr[q][2 * e], (float)xr[q][2 * e + 1]);
    if (s + 2 < nsteps) {
      read_ops(s + 2);                       // po[] currently holds step s+2
      if (s + 4 < nsteps) read_po(s + 4);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int q = 0; q < RQ; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q].v, bp[pl].v, acc[q], 0, 0, 0);
  }

  // ---- combine the two pixel-parity halves, bias gradient, store the slab
  bsum += __shfl_xor(bsum, 32, 64);
  XT_TL(3);
  __syncthreads();                         // dys / pixoff are dead: their LDS is reused for the reduction
  if (kq == 0 && h == 0) bred[pg * 32 + il] = bsum;
  if (pg == 1) {
#pragma unroll
    for (int q = 0; q < RQ; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((kq * RQ + q) * 16 + r) * 64 + lane] = acc[q][r];
  }
  __syncthreads();
  if (pg == 0) {
    const float db = bred[il] + bred[32 + il];
    float* slab = p.out + (size_t)b * ((size_t)(p.KH * 32 + 1) * 32);
    const float corr = p.xb * db;            // d/dW of the (x*xs + xb) transform: xb * sum_p dY
#pragma unroll
    for (int q = 0; q < RQ; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[q][r] + red[((kq * RQ + q) * 16 + r) * 64 + lane];
        const int k = (kq * RQ + q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        store1_wt(slab, (size_t)k * 32 + il, fmaf(v, p.xs, corr));
      }
    if (kq == 0 && h == 0) slab[(size_t)p.KH * 32 * 32 + il] = db;
  }
  XT_TL(4);
  XT_TL_DRAIN(5);
}

// ---- position-flattened weight gradient: one workgroup = 512 consecutive positions of the flattened [B*OH*OW]
// range (32 pixel steps of 16) -> ceil(B*OH*OW / 512) equal workgroups, one per CU, and as many partial slabs
// (250 instead of 320 at B = 320).  Wave roles as above (2 pixel parities x 4 kernel-row groups).
//
// The loop of the first form was ISSUE-bound (ISA: ~200 instructions per 16-pixel step and wave for 6 MFMAs = 192
// matrix cycles: 16 ds_read_u8 + 16 address adds + 24 convert/pack for the x operand, and the 3-way split of the
// SAME eight dY values repeated by all four kernel-row waves).  This form feeds the matrix cores with ~40:
//  * dY is split into its three bf16 planes ONCE, while it is staged (global -> registers -> LDS in B-operand
//    order [step][plane][lane] x 16 B), so a step reads its B operands with three ds_read_b128;
//  * the x operand (row = byte kx*4+c of kernel row ky, reduction = 8 consecutive positions) is ONE
//    ds_read_b64_tr_b8 per kernel row: in every 16-lane group lanes 2j, 2j+1 supply the address of bytes
//    [cb + 0..7], [cb + 8..15] of position j's kernel row and lane c receives column cb + c of the eight positions
//    (semantics measured on the GPU, tools/tr_probe.hip) -- the transposition the byte gather did by hand;
//  * only the input rows the position range needs are staged, packed back to back (<= 44 KB instead of 3 x 28 KB),
//    which is what makes room for the 96 KB of dY planes.
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ i32x2 lds_read_tr8(const uint8_t* p) {
  return __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(p));
}

__global__ __launch_bounds__(512, 2) void conv_u8c4k8_wgrad_flat_kernel(const C1WgArgs p) {
  constexpr int NKQ = 4, NT = 512, RQ = 2, PB = 512, NIMG = 3, NSTEP = PB / 16, NITEM = NSTEP * 64 / NT;
  extern __shared__ __attribute__((aligned(16))) uint8_t lsm[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int HWC = p.H * p.W * 4, Wrow = p.W * 4, OHOW = p.OH * p.OW;
  const int total = p.B * OHOW;
  const int p0 = blockIdx.x * PB, p1 = min(total, p0 + PB);
  uint8_t* limg = lsm;                                                             // packed input rows, <= img_cap
  uint4* dpl = reinterpret_cast<uint4*>(lsm + p.img_cap);                           // [NSTEP][3 planes][64 lanes] x 16 B
  int* pixoff = reinterpret_cast<int*>(lsm + p.img_cap + NSTEP * 3 * 64 * 16);      // [PB]
  float* bred = reinterpret_cast<float*>(pixoff);       // aliases pixoff after the main loop
  XT_TL(0);
  XT_TL_ROLE(50);
  const int il = lane & 31, h = lane >> 5;
  float bsum = 0.f;
  {
    // dY of this thread's NITEM B-operand slots: slot = (step, k half, column) -> 8 pixels x one column (the 32
    // lanes of a half wave read one 128-byte row per load).  All loads of the block are issued before the first use.
    float dv[NITEM][8];
#pragma unroll
    for (int q = 0; q < NITEM; ++q) {
      const int slot = t + NT * q, pos = p0 + (slot >> 6) * 16 + ((slot >> 5) & 1) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) dv[q][e] = p.dy[(size_t)min(pos + e, p1 - 1) * 32 + il];
    }
    int shift[NIMG];
    stage_position_rows_packed<NIMG, NT, 6>(p.in, p.idx, p.B, HWC, Wrow, OHOW, p.OW, p.S, p.KH, p0, p1, limg, t, shift);
#pragma unroll
    for (int q = 0; q < NITEM; ++q) {
      const int slot = t + NT * q, pos = p0 + (slot >> 6) * 16 + ((slot >> 5) & 1) * 8;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[e] = pos + e < p1 ? dv[q][e] : 0.f; bsum += v[e]; }
      bf16x8 pl3[3];
      split3_regs(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), pl3);
      const int st = slot >> 6, ln = slot & 63;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        BF8 u; u.v = pl3[pl];
        dpl[(st * 3 + pl) * 64 + ln] = make_uint4(u.u[0], u.u[1], u.u[2], u.u[3]);
      }
    }
    const int s0 = p0 / OHOW;
    {
      const int pp = min(p0 + t, p1 - 1);        // PB == NT: one position per thread; tail positions carry dY = 0
      const int sidx = pp / OHOW, rem = pp - sidx * OHOW;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      const int di = sidx - s0;
      pixoff[t] = (di == 0 ? shift[0] : di == 1 ? shift[1] : shift[2]) + (p.S * oy * p.W + p.S * ox) * 4;
    }
  }
  const int pg = wave / NKQ, kq = wave - pg * NKQ;
  f32x16 acc[RQ];
#pragma unroll
  for (int q = 0; q < RQ; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  XT_TL(1);
  __syncthreads();
  XT_TL(2);
  // A-operand role of this lane inside its 16-lane group (see the header): position row j, byte half m
  const int c16 = lane & 15, grp = lane >> 4;
  const int prow = 8 * (grp >> 1) + (c16 >> 1);
  const uint8_t* abase = limg + (kq * RQ) * Wrow + (grp & 1) * 16 + (c16 & 1) * 8;
  int po = pixoff[pg * 16 + prow];
  i32x2 xr[RQ];
  uint4 bq[3];
  auto read_ops = [&](int s) {     // uses po of step s
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) bq[pl] = dpl[(s * 3 + pl) * 64 + lane];
#pragma unroll
    for (int q = 0; q < RQ; ++q) xr[q] = lds_read_tr8(abase + po + q * Wrow);
  };
  read_ops(pg);
  po = pixoff[(pg + 2) * 16 + prow];
  // constant trip count (the tail positions of the last block carry dY = 0 and a clamped pixel offset), unrolled
#pragma unroll
  for (int it = 0; it < NSTEP / 2; ++it) {
    const int s = pg + 2 * it;
    BF8 bp[3];
    bf16x8 av[RQ];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) { bp[pl].u[0] = bq[pl].x; bp[pl].u[1] = bq[pl].y; bp[pl].u[2] = bq[pl].z; bp[pl].u[3] = bq[pl].w; }
#pragma unroll
    for (int q = 0; q < RQ; ++q) av[q] = bytes_to_bf16x8((uint32_t)xr[q].x, (uint32_t)xr[q].y);
    if (it + 1 < NSTEP / 2) {
      read_ops(s + 2);                       // po currently holds step s + 2
      if (it + 2 < NSTEP / 2) po = pixoff[(s + 4) * 16 + prow];
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int q = 0; q < RQ; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q], bp[pl].v, acc[q], 0, 0, 0);
  }
  XT_TL(3);
  __syncthreads();                         // every staged operand is dead: the LDS is reused for the combine
  // both pixel-parity halves park their tiles as [parity][k row][36]; then all 512 threads add the two copies and
  // store 16 bytes each: the slab is written as 4 fully coalesced 8 KB passes (the dword form, stored by the four
  // parity-0 waves only, took 3.7 us of the block's 13.9)
  float* T = reinterpret_cast<float*>(lsm);
  bred[(t >> 5) * 32 + il] = bsum;          // 16 row groups x 32 columns of bias-gradient partials (staging threads)
#pragma unroll
  for (int q = 0; q < RQ; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int krow = (kq * RQ + q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      T[(pg * 256 + krow) * 36 + il] = acc[q][r];
    }
  __syncthreads();
  {
    float* slab = p.out + (size_t)blockIdx.x * ((size_t)(p.KH * 32 + 1) * 32);
    const int c4 = (t & 7) * 4;
    float db[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) sum += bred[g * 32 + c4 + c];
      db[c] = sum;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int krow = (t >> 3) + 64 * j;
      const float4 u0 = *reinterpret_cast<const float4*>(&T[krow * 36 + c4]);
      const float4 u1 = *reinterpret_cast<const float4*>(&T[(256 + krow) * 36 + c4]);
      float4 v;                                // d/dW of (x*xs + xb): xb * sum_p dY
      v.x = fmaf(u0.x + u1.x, p.xs, p.xb * db[0]); v.y = fmaf(u0.y + u1.y, p.xs, p.xb * db[1]);
      v.z = fmaf(u0.z + u1.z, p.xs, p.xb * db[2]); v.w = fmaf(u0.w + u1.w, p.xs, p.xb * db[3]);
      *reinterpret_cast<float4*>(slab + (size_t)krow * 32 + c4) = v;
    }
    if (t < 8) *reinterpret_cast<float4*>(slab + (size_t)p.KH * 32 * 32 + c4) = make_float4(db[0], db[1], db[2], db[3]);
  }
  XT_TL(4);
  XT_TL_DRAIN(5);
}

// ================================================================================================================
// The same two kernels for the first layers of ImpalaCnnOpt (xt/model/impala/impala_cnn_opt.py:118-121,
// atari_model.py:4-23): uint8 NHWC C = 4 -> NOUT = 16 channels, square KW x KW kernels at stride KW / 2 (8x8/4 on
// 84x84 frames, 4x4/2 on 42x42), TF SAME padding.  They ran on the generic fp32 implicit-GEMM kernels before -- half
// of every 32-column tile empty, 28-30 % of both IMPALA workloads.  Differences to the PpoCnn forms above:
//  * SAME padding is made PHYSICAL in LDS: the staged image is the zero-padded one (row stride Wp = W + PL + PR pixels,
//    zero rows above / below), so the loops are those of a VALID convolution on it -- no per-operand masks.  The region
//    is zero-filled, then the real pixels are copied in as 8-byte pixel pairs (source rows are 8-byte aligned for even
//    W; the destination is only pixel-aligned because of the left pad: two dword / two 8-byte LDS writes per pair);
//  * NOUT = 16: the B operand (weights / dY) carries zeros in columns 16..31 of the 32x32x16 MFMA -- the bf16 matrix
//    time is irrelevant here -- and the epilogues write NOUT-wide rows;
//  * KW = 4: one 16-byte kernel row = one 16-deep step (the forward) and a 32-row k tile = TWO kernel rows (the weight
//    gradient: the two 16-lane halves of a tr_b8 read address consecutive kernel rows instead of byte halves).
struct C1sArgs {
  const uint8_t* in;
  const int32_t* idx;
  const float* w;      // [KW*KW*4][NOUT]
  const float* bias;   // [NOUT]
  float* y;            // forward: [B*OH*OW][NOUT]
  const float* dy;     // weight gradient: [B*OH*OW][NOUT] d(pre-activation)
  float* out;          // weight gradient: [blocks][(K+1)*NOUT] partial slabs
  int B, H, W, OH, OW, S, PT, PL, Wp, act;
  float xs, mean;      // input transform (x - mean) * xs; mean is an integer in [0, 255]: (x - mean) is then exact in
                       // bf16 (forward: the staged image holds x - mean, pads 0) and a pad pixel of value `mean` is a zero
                       // of the normalised input (weight gradient: uint8 image with pads = mean, the constant term
                       // -mean * xs * sum_p dY[p, n] is added to every row of the slab)
  int img_cap;         // LDS bytes of the staged (padded) image: u8 form; the forward's bf16 image takes twice that
};

// Stage the zero-padded input rows of the position range [p0, p1): stack i's padded rows [ra, rb) go to LDS back to
// back; shift[i] + (rp * Wp + cp) * 4 (+ channel) is the (u8-equivalent) LDS byte of padded pixel (rp, cp) of stack i;
// BF16: LDS byte = 2 * that.  All global loads are issued before the zero fill and its barrier.
template <int NIMG, int NT, int U, bool BF16>
__device__ __forceinline__ void stage_rows_padded(const C1sArgs& p, int KH, int p0, int p1, uint8_t* limg, int t,
                                                  int (&shift)[NIMG]) {
  const int OHOW = p.OH * p.OW, HWC = p.H * p.W * 4, Wp4 = p.Wp * 4, W2 = p.W >> 1;
  const int s0 = p0 / OHOW, slast = (p1 - 1) / OHOW;
  int npair[NIMG], dst0[NIMG], srow[NIMG];
  long long gbase[NIMG];
  int seg = 0, ntot = 0;
#pragma unroll
  for (int i = 0; i < NIMG; ++i) {
    const int sc = min(s0 + i, p.B - 1);
    srow[i] = p.idx ? p.idx[sc] : sc;
  }
#pragma unroll
  for (int i = 0; i < NIMG; ++i) {
    const int sidx = s0 + i;
    npair[i] = 0; dst0[i] = 0; gbase[i] = 0; shift[i] = 0;
    if (sidx <= slast) {
      const int lo = max(p0, sidx * OHOW) - sidx * OHOW, hi = min(p1, (sidx + 1) * OHOW) - 1 - sidx * OHOW;
      const int ra = p.S * (lo / p.OW), rb = p.S * (hi / p.OW) + KH;          // padded rows [ra, rb)
      shift[i] = seg - ra * Wp4;
      const int rlo = max(ra - p.PT, 0), rhi = min(rb - p.PT, p.H);           // image rows inside
      npair[i] = max(rhi - rlo, 0) * W2;
      gbase[i] = (long long)srow[i] * (long long)HWC + (long long)rlo * p.W * 4;
      dst0[i] = shift[i] + ((rlo + p.PT) * p.Wp + p.PL) * 4;
      seg += (rb - ra) * Wp4;
    }
    ntot += npair[i];
  }
  uint2 v[U];
  int dsto[U];
#pragma unroll
  for (int q = 0; q < U; ++q) {
    int u = t + NT * q;
    const bool ok = u < ntot;
    u = ok ? u : 0;
    long long gb = gbase[0];
    int db = dst0[0], ul = u;
#pragma unroll
    for (int j = 0; j + 1 < NIMG; ++j) {
      int cum = 0;
#pragma unroll
      for (int k = 0; k <= j; ++k) cum += npair[k];
      if (u >= cum) { gb = gbase[j + 1]; db = dst0[j + 1]; ul = u - cum; }
    }
    v[q] = *reinterpret_cast<const uint2*>(p.in + gb + (long long)ul * 8);
    const int row = ul / W2, cp = ul - row * W2;
    dsto[q] = ok ? db + (row * p.Wp + 2 * cp) * 4 : -1;
  }
  // pad fill (and everything else), then the pixels
  const int fill16 = ((BF16 ? 2 : 1) * seg + 15) >> 4;
  const uint32_t fw = BF16 ? 0u : (uint32_t)p.mean * 0x01010101u;
  for (int e = t; e < fill16; e += NT) reinterpret_cast<uint4*>(limg)[e] = make_uint4(fw, fw, fw, fw);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < U; ++q) {
    if (dsto[q] >= 0) {
      if constexpr (BF16) {
        BF8 b;
        const uint32_t d0 = v[q].x, d1 = v[q].y;
        const float m = p.mean;
        b.u[0] = pack_hi16((float)(d0 & 0xffu) - m, (float)((d0 >> 8) & 0xffu) - m);
        b.u[1] = pack_hi16((float)((d0 >> 16) & 0xffu) - m, (float)(d0 >> 24) - m);
        b.u[2] = pack_hi16((float)(d1 & 0xffu) - m, (float)((d1 >> 8) & 0xffu) - m);
        b.u[3] = pack_hi16((float)((d1 >> 16) & 0xffu) - m, (float)(d1 >> 24) - m);
        *reinterpret_cast<uint2*>(limg + 2 * dsto[q]) = make_uint2(b.u[0], b.u[1]);
        *reinterpret_cast<uint2*>(limg + 2 * dsto[q] + 8) = make_uint2(b.u[2], b.u[3]);
      } else {
        *reinterpret_cast<uint32_t*>(limg + dsto[q]) = v[q].x;
        *reinterpret_cast<uint32_t*>(limg + dsto[q] + 4) = v[q].y;
      }
    }
  }
}

// padded-image bytes (u8 form) of the largest PB-position range of the launch (host side)
static int c1_padded_cap(int B, int OH, int OW, int S, int KH, int Wp, int PB) {
  const int OHOW = OH * OW, total = B * OHOW;
  int cap = 0;
  for (int p0 = 0; p0 < total; p0 += PB) {
    const int p1 = total < p0 + PB ? total : p0 + PB;
    int bytes = 0;
    for (int sidx = p0 / OHOW; sidx <= (p1 - 1) / OHOW; ++sidx) {
      const int lo = (p0 > sidx * OHOW ? p0 : sidx * OHOW) - sidx * OHOW;
      const int hi = (p1 < (sidx + 1) * OHOW ? p1 : (sidx + 1) * OHOW) - 1 - sidx * OHOW;
      bytes += (S * (hi / OW) + KH - S * (lo / OW)) * Wp * 4;
    }
    if (bytes > cap) cap = bytes;
  }
  return (cap + 15) & ~15;
}

template <int SLOTS, int KW, int NOUT>
__global__ __launch_bounds__(512, (KW == 4 && SLOTS == 1) ? 8 : 2) void conv_u8c4_same_fwd_kernel(const C1sArgs p) {
  constexpr int NW = 8, NT = 512, PB = 32 * NW * SLOTS, NIMG = SLOTS == 2 ? 3 : 2, NS = KW * KW / 4, U = SLOTS == 2 ? 14 : 8;
  extern __shared__ __attribute__((aligned(16))) uint8_t limg[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int OHOW = p.OH * p.OW, WrowP = p.Wp * 4;
  const int total = p.B * OHOW;
  const int p0 = blockIdx.x * PB, p1 = min(total, p0 + PB);
  uint4* wpl = reinterpret_cast<uint4*>(limg + 2 * p.img_cap);     // [NS][3 planes][64 lanes] x 16 B
  XT_TL(0);
  XT_TL_ROLE(40);
  // weights: slot = (step, lane (n = il, k half h)): k = 16 * step + 8 h + j, columns >= NOUT are zero
  constexpr int WQ = (NS * 64 + NT - 1) / NT;
  float wv[WQ][8];
#pragma unroll
  for (int q = 0; q < WQ; ++q) {
    const int slot = t + NT * q;
    const int sc = slot < NS * 64 ? slot : 0;
    const int n = sc & 31, k0 = (sc >> 6) * 16 + 8 * ((sc & 63) >> 5);
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[q][j] = p.w[(size_t)(k0 + j) * NOUT + (n < NOUT ? n : 0)];
  }
  int shift[NIMG];
  stage_rows_padded<NIMG, NT, U, true>(p, KW, p0, p1, limg, t, shift);
  const int s0 = p0 / OHOW;
  XT_TL(1);
#pragma unroll
  for (int q = 0; q < WQ; ++q) {
    const int slot = t + NT * q;
    if (slot < NS * 64) {
      const bool live = (slot & 31) < NOUT;
      BF8 b1, b2, b3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float w0 = live ? wv[q][2 * e] : 0.f, w1 = live ? wv[q][2 * e + 1] : 0.f;
        const float r0 = w0 - trunc_bf16(w0), r1 = w1 - trunc_bf16(w1);
        const float q0 = r0 - trunc_bf16(r0), q1 = r1 - trunc_bf16(r1);
        b1.u[e] = pack_hi16(w0, w1);
        b2.u[e] = pack_hi16(r0, r1);
        b3.u[e] = pack_hi16(q0, q1);
      }
      const int sidx = slot >> 6, ln = slot & 63;
      wpl[(sidx * 3 + 0) * 64 + ln] = make_uint4(b1.u[0], b1.u[1], b1.u[2], b1.u[3]);
      wpl[(sidx * 3 + 1) * 64 + ln] = make_uint4(b2.u[0], b2.u[1], b2.u[2], b2.u[3]);
      wpl[(sidx * 3 + 2) * 64 + ln] = make_uint4(b3.u[0], b3.u[1], b3.u[2], b3.u[3]);
    }
  }
  const int il = lane & 31, h = lane >> 5;
  int poff[SLOTS];                           // LDS byte of this lane's 8 bf16 at step 0 (padded coordinates)
#pragma unroll
  for (int ti = 0; ti < SLOTS; ++ti) {
    const int pp = min(p0 + (wave + NW * ti) * 32 + il, p1 - 1);
    const int sidx = pp / OHOW, rem = pp - sidx * OHOW;
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    const int di = sidx - s0;
    int sh = shift[0];
#pragma unroll
    for (int j = 1; j < NIMG; ++j) sh = di == j ? shift[j] : sh;
    poff[ti] = 2 * (sh + (p.S * oy * p.Wp + p.S * ox) * 4 + 8 * h);
  }
  f32x16 acc[SLOTS];
#pragma unroll
  for (int ti = 0; ti < SLOTS; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;
  const float bias = il < NOUT ? p.bias[il] : 0.f;
  __syncthreads();
  XT_TL(2);
  uint4 wq[2][3], aq[2][SLOTS];
  auto lds_fetch = [&](int s, uint4 (&w3)[3], uint4 (&a2)[SLOTS]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) w3[pl] = wpl[(s * 3 + pl) * 64 + lane];
    const int koff = KW == 8 ? 2 * ((s >> 1) * WrowP + (s & 1) * 16) : 2 * s * WrowP;   // KW = 4: a step = a kernel row
#pragma unroll
    for (int ti = 0; ti < SLOTS; ++ti) a2[ti] = *reinterpret_cast<const uint4*>(limg + poff[ti] + koff);
  };
  lds_fetch(0, wq[0], aq[0]);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int cur = s & 1;
    BF8 bp[3], av[SLOTS];
    if (s + 1 < NS) lds_fetch(s + 1, wq[cur ^ 1], aq[cur ^ 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) { bp[pl].u[0] = wq[cur][pl].x; bp[pl].u[1] = wq[cur][pl].y; bp[pl].u[2] = wq[cur][pl].z; bp[pl].u[3] = wq[cur][pl].w; }
#pragma unroll
    for (int ti = 0; ti < SLOTS; ++ti) { av[ti].u[0] = aq[cur][ti].x; av[ti].u[1] = aq[cur][ti].y; av[ti].u[2] = aq[cur][ti].z; av[ti].u[3] = aq[cur][ti].w; }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int ti = 0; ti < SLOTS; ++ti)
        acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ti].v, bp[pl].v, acc[ti], 0, 0, 0);
  }
  __syncthreads();
  XT_TL(3);
  // epilogue: the tile through LDS (the weight planes are dead) so that every lane stores 16 bytes of a NOUT-wide row
  float* tbuf = reinterpret_cast<float*>(limg) + wave * (32 * 36);
#pragma unroll
  for (int ti = 0; ti < SLOTS; ++ti) {
    const int pix0 = p0 + (wave + NW * ti) * 32;
    if (pix0 < p1) {
      if (p.act == XT_ACT_RELU) {              // un-switched by hand (see the PpoCnn form)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float z = fmaf(acc[ti][r], p.xs, bias);
          tbuf[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + il] = z > 0.f ? z : 0.f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tbuf[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + il] = act_apply(fmaf(acc[ti][r], p.xs, bias), p.act);
      }
      constexpr int LPR = NOUT / 4;            // lanes per output row
#pragma unroll
      for (int q = 0; q < 32 * LPR / 64; ++q) {
        const int row = q * (64 / LPR) + lane / LPR, c4 = (lane % LPR) * 4;
        const float4 v = *reinterpret_cast<const float4*>(&tbuf[row * 36 + c4]);
        if (pix0 + row < p1) *reinterpret_cast<float4*>(&p.y[(size_t)(pix0 + row) * NOUT + c4]) = v;
      }
    }
  }
  XT_TL(4);
  XT_TL_DRAIN(5);
}

// ---- conv1 -> conv2 of ONE frame stack in one workgroup (ImpalaCnnOpt 84x84: 8x8/4 SAME 4 -> 16, then 4x4/2 SAME 16 -> 32;
// xt/model/impala/impala_cnn_opt.py:118-125).  At a few hundred frames per train (examples/breakout_impala.yaml: 128) the
// two forward launches are latency chains of ~7.8 + ~7.0 us that each fill the chip once; here a workgroup runs a whole
// frame: phase 1 = the kernel above on the frame's 441 positions (14 tiles on 8 waves x 2 slots), the activated tiles go to
// global memory (the backward pass reads them) AND stay in LDS; phase 2 = conv2 as an implicit GEMM out of LDS,
// M = 121 positions (4 tiles) x N = 32 x K = 256 on fp32 MFMA (v_mfma_f32_32x32x2_f32): wave = (M tile, K half), the
// wave's 8 kernel taps x 8 channels of the B operand live in 64 registers (loaded while phase 1's epilogue runs), the A
// operand is two 16-byte LDS reads per tap (SAME padding = a zero select).  The K halves are combined through LDS in fixed
// order.  One load phase, one launch boundary and the 1.8 MB round trip of conv1's output less.
struct C1s2Args {
  C1sArgs c1;
  const float* w2;     // [4*4*16][32]
  const float* b2;     // [32]
  float* y2;           // [B*OH2*OW2][32]
  int OH2, OW2, S2, PT2, PL2, act2;
};
constexpr int kA1Stride = 20;      // floats per conv1 output position in LDS (16 channels + 4: 80-byte rows spread the banks)

__global__ __launch_bounds__(512, 1) void conv_u8c4_same_fwd2_kernel(const C1s2Args q2) {
  const C1sArgs& p = q2.c1;
  constexpr int NW = 8, NT = 512, SLOTS = 2, KW = 8, NOUT = 16, NS = KW * KW / 4, U = 14;
  extern __shared__ __attribute__((aligned(16))) uint8_t limg[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int OHOW = p.OH * p.OW, WrowP = p.Wp * 4;
  const int p0 = blockIdx.x * OHOW, p1 = p0 + OHOW;            // one frame stack per workgroup
  uint4* wpl = reinterpret_cast<uint4*>(limg + 2 * p.img_cap);     // [NS][3 planes][64 lanes] x 16 B
  float* a1 = reinterpret_cast<float*>(limg + 2 * p.img_cap + NS * 3 * 64 * 16);     // [OHOW][kA1Stride]
  constexpr int WQ = (NS * 64 + NT - 1) / NT;
  float wv[WQ][8];
#pragma unroll
  for (int q = 0; q < WQ; ++q) {
    const int slot = t + NT * q;
    const int sc = slot < NS * 64 ? slot : 0;
    const int n = sc & 31, k0 = (sc >> 6) * 16 + 8 * ((sc & 63) >> 5);
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[q][j] = p.w[(size_t)(k0 + j) * NOUT + (n < NOUT ? n : 0)];
  }
  int shift[1];
  stage_rows_padded<1, NT, U, true>(p, KW, p0, p1, limg, t, shift);
#pragma unroll
  for (int q = 0; q < WQ; ++q) {
    const int slot = t + NT * q;
    if (slot < NS * 64) {
      const bool live = (slot & 31) < NOUT;
      BF8 b1, b2, b3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float w0 = live ? wv[q][2 * e] : 0.f, w1 = live ? wv[q][2 * e + 1] : 0.f;
        const float r0 = w0 - trunc_bf16(w0), r1 = w1 - trunc_bf16(w1);
        const float q0 = r0 - trunc_bf16(r0), q1 = r1 - trunc_bf16(r1);
        b1.u[e] = pack_hi16(w0, w1);
        b2.u[e] = pack_hi16(r0, r1);
        b3.u[e] = pack_hi16(q0, q1);
      }
      const int sidx = slot >> 6, ln = slot & 63;
      wpl[(sidx * 3 + 0) * 64 + ln] = make_uint4(b1.u[0], b1.u[1], b1.u[2], b1.u[3]);
      wpl[(sidx * 3 + 1) * 64 + ln] = make_uint4(b2.u[0], b2.u[1], b2.u[2], b2.u[3]);
      wpl[(sidx * 3 + 2) * 64 + ln] = make_uint4(b3.u[0], b3.u[1], b3.u[2], b3.u[3]);
    }
  }
  const int il = lane & 31, h = lane >> 5;
  int poff[SLOTS];
#pragma unroll
  for (int ti = 0; ti < SLOTS; ++ti) {
    const int rem = min((wave + NW * ti) * 32 + il, OHOW - 1);
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    poff[ti] = 2 * (shift[0] + (p.S * oy * p.Wp + p.S * ox) * 4 + 8 * h);
  }
  f32x16 acc[SLOTS];
#pragma unroll
  for (int ti = 0; ti < SLOTS; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;
  const float bias = il < NOUT ? p.bias[il] : 0.f;
  __syncthreads();
  uint4 wq[2][3], aq[2][SLOTS];
  auto lds_fetch = [&](int s, uint4 (&w3)[3], uint4 (&a2)[SLOTS]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) w3[pl] = wpl[(s * 3 + pl) * 64 + lane];
    const int koff = 2 * ((s >> 1) * WrowP + (s & 1) * 16);
#pragma unroll
    for (int ti = 0; ti < SLOTS; ++ti) a2[ti] = *reinterpret_cast<const uint4*>(limg + poff[ti] + koff);
  };
  lds_fetch(0, wq[0], aq[0]);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int cur = s & 1;
    BF8 bp[3], av[SLOTS];
    if (s + 1 < NS) lds_fetch(s + 1, wq[cur ^ 1], aq[cur ^ 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) { bp[pl].u[0] = wq[cur][pl].x; bp[pl].u[1] = wq[cur][pl].y; bp[pl].u[2] = wq[cur][pl].z; bp[pl].u[3] = wq[cur][pl].w; }
#pragma unroll
    for (int ti = 0; ti < SLOTS; ++ti) { av[ti].u[0] = aq[cur][ti].x; av[ti].u[1] = aq[cur][ti].y; av[ti].u[2] = aq[cur][ti].z; av[ti].u[3] = aq[cur][ti].w; }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int ti = 0; ti < SLOTS; ++ti)
        acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ti].v, bp[pl].v, acc[ti], 0, 0, 0);
  }
  // phase 2's B operand: wave = (M tile mt, K half kh); taps 8 kh .. 8 kh + 7, lane (n = il, channel half h): 64 values,
  // issued now so that their latency hides behind phase 1's epilogue
  const int mt = wave & 3, kh = wave >> 2;
  float wb[8][8];
#pragma unroll
  for (int tt = 0; tt < 8; ++tt)
#pragma unroll
    for (int j = 0; j < 8; ++j) wb[tt][j] = q2.w2[(size_t)((8 * kh + tt) * 16 + 8 * h + j) * 32 + il];
  const float bias2 = q2.b2[il];
  __syncthreads();
  // phase 1 epilogue: the tile through LDS (the image is dead) -> 16-byte rows to global memory AND into the LDS copy
  float* tbuf = reinterpret_cast<float*>(limg) + wave * (32 * 36);
#pragma unroll
  for (int ti = 0; ti < SLOTS; ++ti) {
    const int row0 = (wave + NW * ti) * 32;              // first position of the tile inside the frame
    if (row0 < OHOW) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tbuf[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + il] = act_apply(fmaf(acc[ti][r], p.xs, bias), p.act);
      constexpr int LPR = NOUT / 4;            // lanes per output row
#pragma unroll
      for (int q = 0; q < 32 * LPR / 64; ++q) {
        const int row = q * (64 / LPR) + lane / LPR, c4 = (lane % LPR) * 4;
        const float4 v = *reinterpret_cast<const float4*>(&tbuf[row * 36 + c4]);
        if (row0 + row < OHOW) {
          *reinterpret_cast<float4*>(&p.y[(size_t)(p0 + row0 + row) * NOUT + c4]) = v;
          *reinterpret_cast<float4*>(&a1[(row0 + row) * kA1Stride + c4]) = v;
        }
      }
    }
  }
  __syncthreads();
  // ---- phase 2: conv2 out of LDS.  Output position pos = 32 mt + il (A rows), taps 8 kh .. 8 kh + 7
  const int OHOW2 = q2.OH2 * q2.OW2;
  const int pos = min(32 * mt + il, OHOW2 - 1);
  const int oy2 = pos / q2.OW2, ox2 = pos - oy2 * q2.OW2;
  f32x16 c2;
#pragma unroll
  for (int r = 0; r < 16; ++r) c2[r] = 0.f;
#pragma unroll
  for (int tt = 0; tt < 8; ++tt) {
    const int tap = 8 * kh + tt, ky = tap >> 2, kx = tap & 3;
    const int iy = q2.S2 * oy2 + ky - q2.PT2, ix = q2.S2 * ox2 + kx - q2.PL2;
    const bool in = iy >= 0 && iy < p.OH && ix >= 0 && ix < p.OW;
    const float* ap = a1 + ((in ? iy : 0) * p.OW + (in ? ix : 0)) * kA1Stride + 8 * h;
    float4 lo = *reinterpret_cast<const float4*>(ap), hi = *reinterpret_cast<const float4*>(ap + 4);
    if (!in) { lo = make_float4(0.f, 0.f, 0.f, 0.f); hi = lo; }
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(lo.x, wb[tt][0], c2, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(lo.y, wb[tt][1], c2, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(lo.z, wb[tt][2], c2, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(lo.w, wb[tt][3], c2, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(hi.x, wb[tt][4], c2, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(hi.y, wb[tt][5], c2, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(hi.z, wb[tt][6], c2, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(hi.w, wb[tt][7], c2, 0, 0, 0);
  }
  // combine the K halves (fixed order: lower half + upper half), bias, activation, 16-byte rows out
  float* cbuf = reinterpret_cast<float*>(limg) + mt * (32 * 36);      // (aliases phase 1's transposition buffers: dead)
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) cbuf[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + il] = c2[r];
  }
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      cbuf[row * 36 + il] = act_apply((c2[r] + cbuf[row * 36 + il]) + bias2, q2.act2);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {               // 32 columns: 8 lanes per row, 8 rows per pass
      const int row = q * 8 + (lane >> 3), c4 = (lane & 7) * 4;
      const float4 v = *reinterpret_cast<const float4*>(&cbuf[row * 36 + c4]);
      if (32 * mt + row < OHOW2)
        *reinterpret_cast<float4*>(&q2.y2[((size_t)blockIdx.x * OHOW2 + 32 * mt + row) * 32 + c4]) = v;
    }
  }
}

// weight gradient: PB positions per workgroup (NSTEP = PB / 16 pixel steps), 8 waves = NPG pixel-step groups x NKQ
// k-tile groups of RQ 32-row k tiles each (K = KW*KW*4: 8 tiles for KW = 8, 2 for KW = 4).
template <int PB, int KW, int NOUT>
__global__ __launch_bounds__(512, 2) void conv_u8c4_same_wgrad_kernel(const C1sArgs p) {
  constexpr int NT = 512, K = KW * KW * 4, NKT = K / 32, NKQ = NKT >= 4 ? 4 : NKT, RQ = NKT / NKQ, NPG = 8 / NKQ;
  constexpr int NIMG = PB == 512 ? 3 : 2, NSTEP = PB / 16, NIT = NSTEP / NPG, U = PB == 512 ? 14 : 8;
  // dY planes hold the NOUT real columns only: [step][plane][k half][NOUT] x 16 B (+ one zero cell that the lanes of the
  // empty columns NOUT..31 read) -- 48 KB instead of 96 KB per 512 positions, i.e. two workgroups per CU
  constexpr int SPP = 2 * NOUT, NITEM = NSTEP * SPP / NT, ZCELL = NSTEP * 3 * SPP;
  static_assert(NITEM >= 1 && NIT >= 1 && NT % NOUT == 0 && NOUT <= 32, "conv1 same wgrad: bad block shape");
  extern __shared__ __attribute__((aligned(16))) uint8_t lsm[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int OHOW = p.OH * p.OW, WrowP = p.Wp * 4;
  const int total = p.B * OHOW;
  const int p0 = blockIdx.x * PB, p1 = min(total, p0 + PB);
  uint8_t* limg = lsm;
  uint4* dpl = reinterpret_cast<uint4*>(lsm + p.img_cap);                           // [NSTEP][3 planes][2][NOUT] x 16 B + zero cell
  int* pixoff = reinterpret_cast<int*>(lsm + p.img_cap + (ZCELL + 1) * 16);          // [PB]
  XT_TL(0);
  XT_TL_ROLE(50);
  const int il = lane & 31, h = lane >> 5;
  float bsum = 0.f;
  {
    // slot = (step, k half, column): 8 pixels x one real column; this thread's column is t % NOUT in all its slots
    float dv[NITEM][8];
    const int col = t % NOUT;
#pragma unroll
    for (int q = 0; q < NITEM; ++q) {
      const int slot = t + NT * q, pos = p0 + (slot / SPP) * 16 + ((slot / NOUT) & 1) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) dv[q][e] = p.dy[(size_t)min(pos + e, p1 - 1) * NOUT + col];
    }
    int shift[NIMG];
    stage_rows_padded<NIMG, NT, U, false>(p, KW, p0, p1, limg, t, shift);
    if (t == 0) dpl[ZCELL] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int q = 0; q < NITEM; ++q) {
      const int slot = t + NT * q, pos = p0 + (slot / SPP) * 16 + ((slot / NOUT) & 1) * 8;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[e] = pos + e < p1 ? dv[q][e] : 0.f; bsum += v[e]; }
      bf16x8 pl3[3];
      split3_regs(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), pl3);
      const int st = slot / SPP, ln = slot % SPP;          // ln = k half * NOUT + column
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        BF8 u; u.v = pl3[pl];
        dpl[(st * 3 + pl) * SPP + ln] = make_uint4(u.u[0], u.u[1], u.u[2], u.u[3]);
      }
    }
    const int s0 = p0 / OHOW;
    if (t < PB) {
      const int pp = min(p0 + t, p1 - 1);
      const int sidx = pp / OHOW, rem = pp - sidx * OHOW;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      const int di = sidx - s0;
      int sh = shift[0];
#pragma unroll
      for (int j = 1; j < NIMG; ++j) sh = di == j ? shift[j] : sh;
      pixoff[t] = sh + (p.S * oy * p.Wp + p.S * ox) * 4;
    }
  }
  const int pg = wave / NKQ, kq = wave - pg * NKQ;
  f32x16 acc[RQ];
#pragma unroll
  for (int q = 0; q < RQ; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  XT_TL(1);
  __syncthreads();
  XT_TL(2);
  // A-operand role of this lane in its 16-lane group: position row j = c16 >> 1 of the 8 positions 8*(grp >> 1) + j,
  // 8-byte half (c16 & 1); the group's 16 columns are bytes (grp & 1)*16.. of a kernel row (KW = 8) or the whole kernel
  // row 2*tile + (grp & 1) (KW = 4)
  const int c16 = lane & 15, grp = lane >> 4;
  const int prow = 8 * (grp >> 1) + (c16 >> 1);
  const uint8_t* abase = limg + (c16 & 1) * 8 +
                         (KW == 8 ? (kq * RQ) * WrowP + (grp & 1) * 16 : (2 * kq * RQ + (grp & 1)) * WrowP);
  constexpr int TSTRIDE_ROWS = KW == 8 ? 1 : 2;      // kernel rows per k tile
  int po = pixoff[pg * 16 + prow];
  i32x2 xr[RQ];
  uint4 bq[3];
  // B operand of lane (column il, k half h): the real columns from the planes, the empty ones from the zero cell (strides 0)
  const uint4* bsrc = dpl + (il < NOUT ? h * NOUT + il : ZCELL);
  const int bstep = il < NOUT ? 3 * SPP : 0, bplane = il < NOUT ? SPP : 0;
  auto read_ops = [&](int s) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) bq[pl] = bsrc[s * bstep + pl * bplane];
#pragma unroll
    for (int q = 0; q < RQ; ++q) xr[q] = lds_read_tr8(abase + po + q * TSTRIDE_ROWS * WrowP);
  };
  read_ops(pg);
  if (NIT > 1) po = pixoff[(pg + NPG) * 16 + prow];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int s = pg + NPG * it;
    BF8 bp[3];
    bf16x8 av[RQ];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) { bp[pl].u[0] = bq[pl].x; bp[pl].u[1] = bq[pl].y; bp[pl].u[2] = bq[pl].z; bp[pl].u[3] = bq[pl].w; }
#pragma unroll
    for (int q = 0; q < RQ; ++q) av[q] = bytes_to_bf16x8((uint32_t)xr[q].x, (uint32_t)xr[q].y);
    if (it + 1 < NIT) {
      read_ops(s + NPG);
      if (it + 2 < NIT) po = pixoff[(s + 2 * NPG) * 16 + prow];
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int q = 0; q < RQ; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q], bp[pl].v, acc[q], 0, 0, 0);
  }
  XT_TL(3);
  __syncthreads();                         // every staged operand is dead: the LDS is reused for the combine
  float* bred = reinterpret_cast<float*>(lsm);         // [16 row groups][32 columns] bias-gradient partials
  float* T = reinterpret_cast<float*>(lsm) + 512;      // [NPG][K rows][36]
  bred[(t / NOUT) * NOUT + t % NOUT] = bsum;          // [NT / NOUT groups][NOUT columns] (= bred[t])
#pragma unroll
  for (int q = 0; q < RQ; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int krow = (kq * RQ + q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      T[(pg * K + krow) * 36 + il] = acc[q][r];
    }
  __syncthreads();
  {
    float* slab = p.out + (size_t)blockIdx.x * ((size_t)(K + 1) * NOUT);
    constexpr int LPR = NOUT / 4;
    for (int e = t; e < K * LPR; e += NT) {
      const int krow = e / LPR, c4 = (e - krow * LPR) * 4;
      float4 v = *reinterpret_cast<const float4*>(&T[krow * 36 + c4]);
#pragma unroll
      for (int g = 1; g < NPG; ++g) {
        const float4 u = *reinterpret_cast<const float4*>(&T[(g * K + krow) * 36 + c4]);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      float db[4];                             // d/dW of ((x - mean) * xs): xs * sum x dY - mean * xs * sum_p dY
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < NT / NOUT; ++g) sum += bred[g * NOUT + c4 + c];
        db[c] = sum;
      }
      const float xb = -p.mean * p.xs;
      v.x = fmaf(v.x, p.xs, xb * db[0]); v.y = fmaf(v.y, p.xs, xb * db[1]);
      v.z = fmaf(v.z, p.xs, xb * db[2]); v.w = fmaf(v.w, p.xs, xb * db[3]);
      *reinterpret_cast<float4*>(slab + (size_t)krow * NOUT + c4) = v;
    }
    if (t < NOUT) {
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < NT / NOUT; ++g) sum += bred[g * NOUT + t];
      slab[(size_t)K * NOUT + t] = sum;
    }
  }
  XT_TL(4);
  XT_TL_DRAIN(5);
}

// geometry test shared by the two launchers below
static bool c1s_geometry(const xt_conv_geom* g, const xt_input_xform* xf) {
  if (!xf || !xf->is_u8 || g->C != 4 || g->N != 16 || g->KH != g->KW || (g->KW != 8 && g->KW != 4)) return false;
  if (g->S * 2 != g->KW || (g->W & 1)) return false;
  const float m = fabsf(xf->mean) >= 1e-4f ? xf->mean : 0.f;      // state_transform: |mean| < 1e-4 -> x / std
  if (m < 0.f || m > 255.f || m != floorf(m)) return false;
  if (g->PT < 0 || g->PL < 0 || g->PT >= g->KH || g->PL >= g->KW) return false;
  return true;
}
static void c1s_fill(C1sArgs* a, const xt_conv_geom* g, const xt_input_xform* xf, int B) {
  a->B = B; a->H = g->H; a->W = g->W; a->OH = g->OH; a->OW = g->OW; a->S = g->S; a->PT = g->PT; a->PL = g->PL;
  const int pr = (g->OW - 1) * g->S + g->KW - g->W - g->PL;        // right pad (SAME: the remainder goes right / below)
  a->Wp = g->W + g->PL + (pr > 0 ? pr : 0);
  a->act = g->act; a->xs = 1.f / xf->std;
  a->mean = fabsf(xf->mean) >= 1e-4f ? xf->mean : 0.f;
}

// returns 0 launched, 1 error, -1 geometry not handled
int launch_conv1_same_fwd(const xt_conv_geom* g, const xt_input_xform* xf, int B, const void* in, const int32_t* idx,
                          const float* w, const float* bias, float* y, hipStream_t st) {
  if (!c1s_geometry(g, xf)) return -1;
  C1sArgs a;
  c1s_fill(&a, g, xf, B);
  a.in = static_cast<const uint8_t*>(in); a.idx = idx; a.w = w; a.bias = bias; a.y = y; a.dy = nullptr; a.out = nullptr;
  const int OHOW = g->OH * g->OW, total = B * OHOW;
  // 4x4 kernels: 256-position ranges always -- 63 VGPRs, four workgroups per CU hide each other's staging latency
  const bool two = g->KW == 8 && (total + 511) / 512 >= 200;
  const int pb = two ? 512 : 256, nimg = two ? 3 : 2;
  if ((pb - 1) / OHOW + 2 > nimg) return -1;
  a.img_cap = c1_padded_cap(B, g->OH, g->OW, g->S, g->KH, a.Wp, pb);
  const int ns = g->KW * g->KW / 4;
  size_t fl = (size_t)2 * a.img_cap + (size_t)ns * 3 * 64 * 16;
  if (fl < (size_t)8 * 32 * 36 * 4) fl = (size_t)8 * 32 * 36 * 4;            // the output transpose aliases the image
  if (fl > 160 * 1024) return -1;
  const int maxpairs = a.img_cap / 8, u = two ? 14 : 8;
  if (maxpairs > 512 * u) return -1;                                          // staging: U pixel pairs per thread
  static PerDeviceOnce attr_once;
  attr_once.run([] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_u8c4_same_fwd_kernel<2, 8, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_u8c4_same_fwd_kernel<1, 8, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_u8c4_same_fwd_kernel<2, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_u8c4_same_fwd_kernel<1, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
  });
  const dim3 grid((total + pb - 1) / pb), blk(512);
  if (g->KW == 8) {
    if (two) hipLaunchKernelGGL((conv_u8c4_same_fwd_kernel<2, 8, 16>), grid, blk, fl, st, a);
    else hipLaunchKernelGGL((conv_u8c4_same_fwd_kernel<1, 8, 16>), grid, blk, fl, st, a);
  } else {
    if (two) hipLaunchKernelGGL((conv_u8c4_same_fwd_kernel<2, 4, 16>), grid, blk, fl, st, a);
    else hipLaunchKernelGGL((conv_u8c4_same_fwd_kernel<1, 4, 16>), grid, blk, fl, st, a);
  }
  XT_LAUNCH_CHECK();
  return 0;
}

// conv1 -> conv2 of a frame stack in one launch (conv_u8c4_same_fwd2_kernel): returns 0 launched, 1 error, -1 geometry not
// handled (the caller then launches the two layers separately)
int launch_conv12_same_fwd(const xt_conv_geom* g, const xt_input_xform* xf, const xt_conv_geom* g2, int B, const void* in,
                           const int32_t* idx, const float* w, const float* bias, float* y, const float* w2, const float* b2,
                           float* y2, hipStream_t st) {
  if (!c1s_geometry(g, xf) || g->KW != 8) return -1;
  const int OHOW = g->OH * g->OW, OHOW2 = g2->OH * g2->OW;
  if (OHOW > 14 * 32 || g2->C != 16 || g2->N != 32 || g2->KH != 4 || g2->KW != 4 || g2->H != g->OH || g2->W != g->OW ||
      OHOW2 > 128 || g2->PT < 0 || g2->PL < 0)
    return -1;
  C1s2Args a;
  c1s_fill(&a.c1, g, xf, B);
  a.c1.in = static_cast<const uint8_t*>(in); a.c1.idx = idx; a.c1.w = w; a.c1.bias = bias; a.c1.y = y; a.c1.dy = nullptr; a.c1.out = nullptr;
  // the whole padded frame: rows [0, S (OH - 1) + KH)
  a.c1.img_cap = ((g->S * (g->OH - 1) + g->KH) * a.c1.Wp * 4 + 15) & ~15;
  a.w2 = w2; a.b2 = b2; a.y2 = y2; a.OH2 = g2->OH; a.OW2 = g2->OW; a.S2 = g2->S; a.PT2 = g2->PT; a.PL2 = g2->PL; a.act2 = g2->act;
  const size_t fl = (size_t)2 * a.c1.img_cap + (size_t)16 * 3 * 64 * 16 + (size_t)OHOW * kA1Stride * 4;
  if (fl > 160 * 1024 || (size_t)2 * a.c1.img_cap < (size_t)8 * 32 * 36 * 4) return -1;
  if (a.c1.img_cap / 8 > 512 * 14) return -1;                                   // staging: 14 pixel pairs per thread
  static PerDeviceOnce attr_once;
  attr_once.run([] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_u8c4_same_fwd2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
  });
  hipLaunchKernelGGL(conv_u8c4_same_fwd2_kernel, dim3(B), dim3(512), fl, st, a);
  XT_LAUNCH_CHECK();
  return 0;
}

int launch_conv1_same_wgrad(const xt_conv_geom* g, const xt_input_xform* xf, int B, const void* in, const int32_t* idx,
                            const float* dy, float* dwb, float* slabs, int max_slabs, int* msplit_out, hipStream_t st) {
  if (!c1s_geometry(g, xf) || !slabs) return -1;
  C1sArgs a;
  c1s_fill(&a, g, xf, B);
  a.in = static_cast<const uint8_t*>(in); a.idx = idx; a.w = nullptr; a.bias = nullptr; a.y = nullptr; a.dy = dy;
  const int OHOW = g->OH * g->OW, total = B * OHOW;
  const bool two = (total + 511) / 512 >= 200;      // (256-position ranges for 4x4 kernels as in the forward: kernel
                                                    //  19.3 -> 17.8 us but twice the slabs: pong 270 -> 280 us per train)
  const int pb = two ? 512 : 256, nimg = two ? 3 : 2, nblk = (total + pb - 1) / pb;
  if ((pb - 1) / OHOW + 2 > nimg || nblk > max_slabs) return -1;
  a.out = nblk == 1 ? dwb : slabs;
  a.img_cap = c1_padded_cap(B, g->OH, g->OW, g->S, g->KH, a.Wp, pb);
  const int K = g->KW * g->KW * 4, npg = g->KW == 8 ? 2 : 4;
  size_t fl = (size_t)a.img_cap + ((size_t)(pb / 16) * 3 * 2 * 16 + 1) * 16 + (size_t)pb * 4;
  if (fl < (size_t)2048 + (size_t)npg * K * 36 * 4) fl = (size_t)2048 + (size_t)npg * K * 36 * 4;   // combine buffers alias everything
  if (fl > 160 * 1024) return -1;
  const int maxpairs = a.img_cap / 8, u = two ? 14 : 8;
  if (maxpairs > 512 * u) return -1;
  static PerDeviceOnce attr_once;
  attr_once.run([] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_u8c4_same_wgrad_kernel<512, 8, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_u8c4_same_wgrad_kernel<256, 8, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_u8c4_same_wgrad_kernel<512, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_u8c4_same_wgrad_kernel<256, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
  });
  const dim3 grid(nblk), blk(512);
  if (g->KW == 8) {
    if (two) hipLaunchKernelGGL((conv_u8c4_same_wgrad_kernel<512, 8, 16>), grid, blk, fl, st, a);
    else hipLaunchKernelGGL((conv_u8c4_same_wgrad_kernel<256, 8, 16>), grid, blk, fl, st, a);
  } else {
    if (two) hipLaunchKernelGGL((conv_u8c4_same_wgrad_kernel<512, 4, 16>), grid, blk, fl, st, a);
    else hipLaunchKernelGGL((conv_u8c4_same_wgrad_kernel<256, 4, 16>), grid, blk, fl, st, a);
  }
  XT_LAUNCH_CHECK();
  if (msplit_out) *msplit_out = nblk;
  return 0;
}

XT_TL_SETTER(conv1)

// returns 0 launched (msplit_out = B slabs), 1 error, -1 geometry not handled
int launch_conv1_wgrad_bf16x3(const xt_conv_geom* g, const xt_input_xform* xf, int B, const void* in,
                              const int32_t* idx, const float* dy, float* dwb, float* slabs, int max_slabs,
                              int* msplit_out, hipStream_t st) {
  if (!xf || !xf->is_u8 || g->C != 4 || g->KW != 8 || g->KH != 8 || g->N != 32 || g->PT != 0 || g->PL != 0) return -1;
  if ((g->OH - 1) * g->S + g->KH > g->H || (g->OW - 1) * g->S + g->KW > g->W) return -1;
  const int HWC = g->H * g->W * 4;
  if (HWC % 16 != 0 || HWC > 64 * 1024) return -1;
  if (!slabs || B > max_slabs) return -1;
  C1WgArgs a;
  a.in = static_cast<const uint8_t*>(in); a.idx = idx; a.dy = dy; a.out = (B == 1) ? dwb : slabs;
  a.B = B; a.H = g->H; a.W = g->W; a.OH = g->OH; a.OW = g->OW; a.S = g->S; a.KH = g->KH; a.img_cap = 0;
  const float mean = fabsf(xf->mean) >= 1e-4f ? xf->mean : 0.f;
  a.xs = 1.f / xf->std; a.xb = -mean * a.xs;
  const int nsteps = (g->OH * g->OW + 15) / 16;
  if (nsteps * 16 * 8 > 16 * 256) return -1;                 // dY staging: 16 float4 per thread
  size_t lds = (size_t)HWC + (size_t)nsteps * 16 * 32 * 4 + (size_t)nsteps * 16 * 4;
  if ((size_t)nsteps * 16 * 32 * 4 < (size_t)2 * 4 * 16 * 64 * 4) return -1;   // `red` aliases the dY region
  if ((size_t)nsteps * 16 * 4 < 64 * 4) return -1;                              // `bred` aliases pixoff
  if (lds > 81920) return -1;                                                    // two workgroups per CU
  {
    const int flat = tuning().conv1_flat;
    const int total = B * g->OH * g->OW, nblk = (total + 511) / 512;
    a.img_cap = c1_packed_cap(B, g->OH, g->OW, g->S, g->KH, g->W * 4, 512);
    size_t fl = (size_t)a.img_cap + (size_t)32 * 3 * 64 * 16 + (size_t)512 * 4;
    if (fl < (size_t)2 * 256 * 36 * 4) fl = (size_t)2 * 256 * 36 * 4;            // the combine buffer aliases everything
    if (flat && c1_waves() == 8 && nblk >= 200 && nblk <= max_slabs && 511 / (g->OH * g->OW) + 2 <= 3 &&
        fl <= 160 * 1024 && B > 1 && (g->W * 4) % 8 == 0 && (g->S * 4) % 16 == 0) {
      static PerDeviceOnce attr_once;
      attr_once.run([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_u8c4k8_wgrad_flat_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
      });
      hipLaunchKernelGGL(conv_u8c4k8_wgrad_flat_kernel, dim3(nblk), dim3(512), fl, st, a);
      XT_LAUNCH_CHECK();
      if (msplit_out) *msplit_out = nblk;
      return 0;
    }
  }
  if (c1_waves() == 8) hipLaunchKernelGGL(conv_u8c4k8_wgrad_bf16x3_kernel<4>, dim3(B), dim3(512), lds, st, a);
  else hipLaunchKernelGGL(conv_u8c4k8_wgrad_bf16x3_kernel<2>, dim3(B), dim3(256), lds, st, a);
  XT_LAUNCH_CHECK();
  if (msplit_out) *msplit_out = B;
  return 0;
}

}  // namespace xt
