// Fused forward of the PpoCnn convolution trunk: conv1 (uint8 8x8/4, bf16x3) -> conv2 -> conv3 of ONE frame
// stack per workgroup, intermediate activations kept in LDS.
//
// Why (profiles/r01_timeline.txt): as three launches the trunk forward costs 19 + 17 + 12 us for 0.37 GFLOP-class
// GEMMs; the im2col gathers of conv2/conv3 go through L2->L1 at ~25 B/clk/CU with a 4x/9x re-read of every input
// element, each launch pays its own dependent prologue/epilogue and the ~2 us launch-to-launch gap.  Per frame
// stack the intermediate tensors are tiny (act1 51 KB, act2 10 KB), so one workgroup can run the whole stack:
//   phase 0  frame stack (28 KB uint8) -> LDS                                  (coalesced 16 B loads, gather fused)
//   phase 1  conv1 on v_mfma_f32_32x32x16_bf16, exact 3-way weight split done in registers (weights straight
//            from L2 in MFMA operand order) -> act1 into LDS, channel-major [c][stride-parity plane][iy][ix]
//   phase 2  conv2 on v_mfma_f32_32x32x2_f32: A operand = 16 ds_read_b32 per step from act1 (lanes = pixels ->
//            consecutive words, conflict-free; odd channel stride), B operand = weights from L2 (32 consecutive
//            output channels per load); waves <-> 32-pixel tiles; the wave(s) without a tile stream act1 to HBM
//            (the backward pass needs it) meanwhile.  act2 -> HBM + LDS (the dead frame-stack region)
//   phase 3  conv3 likewise from act2 -> act3 to HBM (the dense layer's input)
// HBM traffic of the trunk forward drops to the compulsory reads/writes; every MFMA operand gather hits LDS.
// Same arithmetic as the per-layer kernels (fp32 products and accumulation; conv1 exact bf16x3), only the
// summation order inside a dot product differs.
//
// Envelope (else the caller runs the per-layer kernels): layer 0 as xt_conv1.hip (uint8, C=4, KW=8, N=32, VALID,
// mean 0); layer 1: C=32 -> N=32, VALID, any kernel/stride; layer 2: C=32 -> N=32 or 64, VALID, stride 1;
// at most 2 tiles per wave and stage; LDS <= 80 KB per workgroup (two workgroups per CU).
#include <math.h>
#include "xt_common.h"
#include "xt_conv1_dev.h"
#include "xt_direct_dev.h"

namespace xt {

struct TrunkFwdArgs {
  const uint8_t* in;
  const int32_t* idx;
  // layer 0
  const float* w1; const float* b1; float* y1;
  int H, W, OH1, OW1, S1, KH1, act1;
  float xs;
  // layer 1
  const float* w2; const float* b2; float* y2;
  int KH2, KW2, S2, OH2, OW2, act2;
  int PW1, PHW1, CS1;          // act1 in LDS: plane width, plane size, channel stride (odd)
  FastDiv d_ow1, d_s2, d_ow2;
  // layer 2
  const float* w3; const float* b3; float* y3;
  int KH3, KW3, OH3, OW3, N3, act3;
  int CS2;                     // act2 in LDS: channel stride (odd), row pitch = OW2
  FastDiv d_ow3;
  int B;
  int off_act1;                // LDS byte offset of the act1 region (the frame stack / act2 region starts at 0)
  int off_tab;                 // LDS byte offset of the index tables
};

// LDS word index of conv1 output pixel p inside one channel of act1
__device__ __forceinline__ int act1_index(const TrunkFwdArgs& p, int pix) {
  const uint32_t y = fdiv((uint32_t)pix, p.d_ow1), x = (uint32_t)pix - y * (uint32_t)p.OW1;
  const uint32_t iy = fdiv(y, p.d_s2), ix = fdiv(x, p.d_s2);
  const uint32_t py = y - iy * (uint32_t)p.S2, px = x - ix * (uint32_t)p.S2;
  return (int)((py * (uint32_t)p.S2 + px) * (uint32_t)p.PHW1 + iy * (uint32_t)p.PW1 + ix);
}

// Partial 32x32 output tile over taps [t0, t1) of a VALID conv whose 32 input channels sit channel-major in LDS.
//   act: LDS base of the input [32][CS]; lane_off: this lane's pixel word offset at tap (0,0); taptab[t]: word
//   offset of tap t (LDS table, uniform read); weights [ntaps*32][N] in HBM/L2.  CSC > 0: compile-time channel
//   stride (the 16 per-step ds_read_b32 then share ONE address register, kk*CS goes into the offset field).
// The weight operand comes from L2 with ~2000 cycles of latency when every workgroup of the chip reads the same
// few KB (measured: a one-step prefetch left the loop latency-bound at 3x its MFMA time), so it is fetched TWO
// steps ahead and the first two steps are issued by prefetch(), which the caller places before the barrier
// that publishes the input activations (with two waves per SIMD two steps are ~4000 cycles).  The LDS operand is fetched one
// step ahead.  Two accumulators (even/odd reduction index).
template <int CSC>
struct LdsConvPart {
  float b0[16], b1[16];
  __amdgpu_buffer_rsrc_t rs;
  uint32_t wvoff;
  int N, t0, t1;

  __device__ __forceinline__ void loadB(float (&b)[16], int t) {
    const bool live = t < t1;
    const uint32_t ws = (uint32_t)((live ? t : t0) * 32 * N) * 4u;
    const uint32_t vo = live ? wvoff : kOob;     // dead: out of range through the LANE offset -> zeros, no access
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) b[kk] = buf_load1(rs, vo, ws + (uint32_t)(kk * N) * 4u);
  }
  __device__ __forceinline__ void prefetch(__amdgpu_buffer_rsrc_t rs_w, int N_, int n0, int lane, int t0_, int t1_) {
    rs = rs_w; N = N_; t0 = t0_; t1 = t1_;
    wvoff = (uint32_t)(16 * (lane >> 5) * N + n0 + (lane & 31)) * 4u;
    loadB(b0, t0); loadB(b1, t0 + 1);
  }
  __device__ __forceinline__ void run(const float* act, int CSr, int lane_off, const int* taptab, int lane, f32x16& out) {
    const int CS = CSC > 0 ? CSC : CSr;
    const float* abase = act + (16 * (lane >> 5)) * CS + lane_off;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float a0[16], a1[16];
    auto loadA = [&](float (&a)[16], int t) {
      const float* ap = abase + taptab[t < t1 ? t : t0];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) a[kk] = ap[kk * CS];
    };
    auto compute = [&](const float (&a)[16], const float (&b)[16]) {
#pragma unroll
      for (int kk = 0; kk < 16; kk += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk + 1], b[kk + 1], acc1, 0, 0, 0);
      }
    };
    loadA(a0, t0);
#define XT_STEP(AC, AN, BX, T)          \
    loadA(AN, (T) + 1);                 \
    compute(AC, BX);                    \
    loadB(BX, (T) + 2);
    for (int t = t0; t < t1; t += 2) {
      XT_STEP(a0, a1, b0, t)
      if (t + 1 >= t1) break;
      XT_STEP(a1, a0, b1, t + 1)
    }
#undef XT_STEP
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r] = acc0[r] + acc1[r];
  }
};

constexpr int kTW = 8;              // waves per workgroup: two per SIMD, so one wave's operand fetch / address /
                                    // epilogue phases run in the shadow of the other's MFMAs (a lone wave issues
                                    // in order: measured 153 instead of 64 cycles per fp32 MFMA)
constexpr int kTrunkTiles1 = 2;     // conv1: 32-pixel tiles per wave (OH1*OW1 <= 512)

template <int CS1C, int CS2C>
__global__ __launch_bounds__(64 * kTW, 4) void trunk_fwd_kernel(const TrunkFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int il = lane & 31, h = lane >> 5;
  const int b = blockIdx.x;
  const int HWC = p.H * p.W * 4, Wrow = p.W * 4;
  const int M1 = p.OH1 * p.OW1, M2 = p.OH2 * p.OW2, M3 = p.OH3 * p.OW3;
  const int CS1 = CS1C > 0 ? CS1C : p.CS1, CS2 = CS2C > 0 ? CS2C : p.CS2;
  uint8_t* limg = lds;                                              // phase 0/1
  float* act2 = reinterpret_cast<float*>(lds);                      // phase 2/3 (the frame stack is dead by then)
  float* red2 = act2 + 32 * CS2;                                    // phase 2: K-split partials of conv2
  float* act1 = reinterpret_cast<float*>(lds + p.off_act1);
  float* red3 = act1;                                               // phase 3: K-split partials of conv3 (act1 is dead)
  uint16_t* tab1 = reinterpret_cast<uint16_t*>(lds + p.off_tab);   // conv1 pixel -> word index inside an act1 channel
  int* taps2 = reinterpret_cast<int*>(lds + p.off_tab + ((M1 * 2 + 15) & ~15));
  int* taps3 = taps2 + p.KH2 * p.KW2;
  XT_TL(0);
  XT_TL_ROLE(90);

  // ---------------- phase 0: frame stack -> LDS (minibatch gather fused), index tables
  {
    const size_t s = p.idx ? (size_t)p.idx[b] : (size_t)b;
    const uint4* src = reinterpret_cast<const uint4*>(p.in + s * (size_t)HWC);
    uint4* dst = reinterpret_cast<uint4*>(limg);
    const int n16 = HWC >> 4;
    for (int base = 0; base < n16; base += 64 * kTW * 4) {
      uint4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int i = base + t + 64 * kTW * q; v[q] = src[i < n16 ? i : 0]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int i = base + t + 64 * kTW * q; if (i < n16) dst[i] = v[q]; }
    }
    for (int pix = t; pix < M1; pix += 64 * kTW) tab1[pix] = (uint16_t)act1_index(p, pix);
    if (t < p.KH2 * p.KW2) {
      const int ky = t / p.KW2, kx = t - ky * p.KW2;
      const int qy = ky / p.S2, qx = kx / p.S2;
      taps2[t] = ((ky - qy * p.S2) * p.S2 + (kx - qx * p.S2)) * p.PHW1 + qy * p.PW1 + qx;
    }
    if (t < p.KH3 * p.KW3) {
      const int ky = t / p.KW3, kx = t - ky * p.KW3;
      taps3[t] = ky * p.OW2 + kx;
    }
  }
  const int nsteps1 = 2 * p.KH1;
  const int ntiles1 = (M1 + 31) >> 5;
  int poff[kTrunkTiles1];
#pragma unroll
  for (int ti = 0; ti < kTrunkTiles1; ++ti) {
    const int pix = min((wave + kTW * ti) * 32 + il, M1 - 1);
    const uint32_t oy = fdiv((uint32_t)pix, p.d_ow1), ox = (uint32_t)pix - oy * (uint32_t)p.OW1;
    poff[ti] = (p.S1 * (int)oy * p.W + p.S1 * (int)ox) * 4 + 8 * h;
  }
  // ---------------- phase 1: conv1, exact bf16x3
  f32x16 acc[kTrunkTiles1];
#pragma unroll
  for (int ti = 0; ti < kTrunkTiles1; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;
  // weights: exact 3-way bf16 split, written once per workgroup in MFMA operand order into the (still unused)
  // act1 region: [nsteps1][3 planes][64 lanes] x 16 B.  (Fetching them per step from L2 instead left this phase
  // latency-bound: every workgroup of the chip reads the same 32 KB at the same time.)
  uint4* wpl = reinterpret_cast<uint4*>(act1);
  {
    const int nslots = nsteps1 * 64;
    float wv[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int slot = t + 64 * kTW * q;
      const int sc = slot < nslots ? slot : 0;
      const float* wl = p.w1 + (size_t)((sc >> 6) * 16 + 8 * ((sc & 63) >> 5)) * 32 + (sc & 31);
#pragma unroll
      for (int j = 0; j < 8; ++j) wv[q][j] = wl[j * 32];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int slot = t + 64 * kTW * q;
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
        const int sp = slot >> 6, ln = slot & 63;
        wpl[(sp * 3 + 0) * 64 + ln] = make_uint4(b1.u[0], b1.u[1], b1.u[2], b1.u[3]);
        wpl[(sp * 3 + 1) * 64 + ln] = make_uint4(b2.u[0], b2.u[1], b2.u[2], b2.u[3]);
        wpl[(sp * 3 + 2) * 64 + ln] = make_uint4(b3.u[0], b3.u[1], b3.u[2], b3.u[3]);
      }
    }
  }
  const int ntiles2 = (M2 + 31) >> 5;
  const int ntaps2 = p.KH2 * p.KW2, thalf2 = (ntaps2 + 1) >> 1;
  const int tile2 = wave % ntiles2, half2 = wave / ntiles2;      // conv2 item = wave (2*ntiles2 <= kTW, host)
  LdsConvPart<CS1C> c2;
  __syncthreads();                        // frame stack, weight planes and tables staged
  XT_TL(1);
  const bool two1 = wave + kTW < ntiles1;      // wave-uniform: does this wave own a second tile?
  if (wave < ntiles1) {
    uint4 wq[3];
    uint2 aq[kTrunkTiles1];
    auto lds_fetch = [&](int s) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) wq[pl] = wpl[(s * 3 + pl) * 64 + lane];
      const int koff = (s >> 1) * Wrow + (s & 1) * 16;
#pragma unroll
      for (int ti = 0; ti < kTrunkTiles1; ++ti) aq[ti] = *reinterpret_cast<const uint2*>(limg + poff[ti] + koff);
    };
    lds_fetch(0);
    for (int s = 0; s < nsteps1; ++s) {
      BF8 bp[3];
      bf16x8 av[kTrunkTiles1];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) { bp[pl].u[0] = wq[pl].x; bp[pl].u[1] = wq[pl].y; bp[pl].u[2] = wq[pl].z; bp[pl].u[3] = wq[pl].w; }
#pragma unroll
      for (int ti = 0; ti < kTrunkTiles1; ++ti) av[ti] = bytes_to_bf16x8(aq[ti].x, aq[ti].y);
      if (s + 1 < nsteps1) lds_fetch(s + 1);
      // unconditional MFMAs (a wave without a second tile computes a duplicate of pixel M1-1 and drops it): a
      // branch around every MFMA breaks their back-to-back issue
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int ti = 0; ti < kTrunkTiles1; ++ti)
          acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ti], bp[pl].v, acc[ti], 0, 0, 0);
    }
  }
  if (wave < 2 * ntiles2)                 // conv2's first weight steps only depend on the parameters: request them now
    c2.prefetch(make_rsrc(p.w2, (uint32_t)(ntaps2 * 32 * 32) * 4u), 32, 0, lane, half2 ? thalf2 : 0, half2 ? ntaps2 : thalf2);
  __syncthreads();                        // every wave is done with the weight planes: act1 may overwrite them
  if (wave < ntiles1) {
    const float bias = p.b1[il];
#pragma unroll
    for (int ti = 0; ti < kTrunkTiles1; ++ti) {
      if (ti == 0 || two1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pix = (wave + kTW * ti) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (pix < M1) act1[il * CS1 + tab1[pix]] = act_apply(fmaf(acc[ti][r], p.xs, bias), p.act1);
        }
      }
    }
  }
  // conv3's first weight steps likewise (they are consumed after two more barriers)
  const int nt3 = p.N3 >> 5;
  const int ntiles3 = ((M3 + 31) >> 5) * nt3;
  const int ntaps3 = p.KH3 * p.KW3, thalf3 = (ntaps3 + 1) >> 1;
  const int tile3 = wave % ntiles3, half3 = wave / ntiles3;
  const int mt3 = tile3 / nt3, n03 = (tile3 - mt3 * nt3) * 32;
  __syncthreads();                        // act1 complete; the frame stack region is free
  XT_TL(2);

  // ---------------- phase 2: conv2 from LDS, (tile, reduction half) per wave; idle waves stream act1 to HBM
  auto act1_out = [&](int part, int nparts) {      // NHWC copy-out: 8 lanes x 16 B = one pixel's 32 channels
    float* y = p.y1 + (size_t)b * M1 * 32;
    const int c4 = (lane & 7) * 4;
    for (int pix = part * 8 + (lane >> 3); pix < M1; pix += nparts * 8) {
      const int li = tab1[pix];
      float4 v;
      v.x = act1[(c4 + 0) * CS1 + li]; v.y = act1[(c4 + 1) * CS1 + li];
      v.z = act1[(c4 + 2) * CS1 + li]; v.w = act1[(c4 + 3) * CS1 + li];
      *reinterpret_cast<float4*>(y + (size_t)pix * 32 + c4) = v;
    }
  };
  LdsConvPart<CS2C> c3;
  {
    const int nitems = 2 * ntiles2;
    if (nitems >= kTW) act1_out(wave, kTW);          // nobody idles in phase 2: everybody copies first
    f32x16 o;
    if (wave < nitems) {
      const int m = min(tile2 * 32 + il, M2 - 1);
      const uint32_t oy = fdiv((uint32_t)m, p.d_ow2), ox = (uint32_t)m - oy * (uint32_t)p.OW2;
      c2.run(act1, CS1, (int)oy * p.PW1 + (int)ox, taps2, lane, o);
      if (half2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red2[(tile2 * 16 + r) * 64 + lane] = o[r];
      }
    } else if (nitems < kTW) {
      act1_out(wave - nitems, kTW - nitems);
    }
    if (wave < 2 * ntiles3)
      c3.prefetch(make_rsrc(p.w3, (uint32_t)(ntaps3 * 32 * p.N3) * 4u), p.N3, n03, lane, half3 ? thalf3 : 0,
                  half3 ? ntaps3 : thalf3);
    __syncthreads();                      // reduction halves meet
    if (wave < nitems && !half2) {
      const float bias = p.b2[il];
      float* y = p.y2 + (size_t)b * M2 * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mm = tile2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (mm < M2) {
          const float v = act_apply(o[r] + red2[(tile2 * 16 + r) * 64 + lane] + bias, p.act2);
          y[(size_t)mm * 32 + il] = v;
          act2[il * CS2 + mm] = v;
        }
      }
    }
  }
  __syncthreads();                        // act2 complete
  XT_TL(3);

  // ---------------- phase 3: conv3 from LDS -> HBM, (tile, reduction half) per wave
  {
    const int nitems = 2 * ntiles3;
    f32x16 o;
    if (wave < nitems) {
      const int m = min(mt3 * 32 + il, M3 - 1);
      const uint32_t oy = fdiv((uint32_t)m, p.d_ow3), ox = (uint32_t)m - oy * (uint32_t)p.OW3;
      c3.run(act2, CS2, (int)oy * p.OW2 + (int)ox, taps3, lane, o);
      if (half3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red3[(tile3 * 16 + r) * 64 + lane] = o[r];
      }
    }
    __syncthreads();
    if (wave < nitems && !half3) {
      const float bias = p.b3[n03 + il];
      float* y = p.y3 + (size_t)b * M3 * p.N3;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mm = mt3 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (mm < M3) y[(size_t)mm * p.N3 + n03 + il] = act_apply(o[r] + red3[(tile3 * 16 + r) * 64 + lane] + bias, p.act3);
      }
    }
  }
  XT_TL(4);
  XT_TL_DRAIN(5);
}

XT_TL_SETTER(trunk)

// Opt-in (XT_TRUNK=1, read on every call so that tests can toggle it): measured on MI355X at B=320 the fused
// kernel takes 50.4 us against 48.2 us for the three per-layer launches it replaces (profiles/r01_timeline_trunk.txt).
// Alone on a CU a frame stack takes 31.6 us (conv1 9.2 / conv2 11.7 / conv3 7.8, all latency-bound: MFMA pipe 29 %
// busy, SQ_WAIT_INST_ANY 40 % of the wave cycles), but 80 KB of LDS allow only two workgroups per CU, so the 64 CUs
// that get a second frame stack (320 on 256 CUs) finish at 46 us and set the kernel time.  What it already
// delivers: HBM traffic of the trunk forward at the compulsory minimum and every conv2/conv3 operand gather from
// LDS.  Open: a finer-than-frame-stack work split for the 64 extra samples.
static bool use_trunk() {
  const char* e = getenv("XT_TRUNK");
  return e && e[0] == '1';
}

// returns -1 when the three layers are outside the envelope (the caller runs them one by one)
int launch_trunk_fwd(const xt_conv_geom* g1, const xt_conv_geom* g2, const xt_conv_geom* g3, const xt_input_xform* xf,
                     int B, const void* in, const int32_t* idx, const float* w1, const float* b1, float* y1,
                     const float* w2, const float* b2, float* y2, const float* w3, const float* b3, float* y3,
                     hipStream_t st, bool force) {
  if (!force && !use_trunk()) return -1;
  // layer 0: the xt_conv1.hip envelope
  if (!xf || !xf->is_u8 || g1->C != 4 || g1->KW != 8 || g1->N != 32 || g1->PT != 0 || g1->PL != 0) return -1;
  if ((g1->OH - 1) * g1->S + g1->KH > g1->H || (g1->OW - 1) * g1->S + g1->KW > g1->W) return -1;
  const int HWC = g1->H * g1->W * 4;
  if (HWC % 16 != 0 || (g1->W * 4) % 8 != 0 || (g1->S * 4) % 8 != 0 || g1->KH > 8) return -1;
  if (fabsf(xf->mean) >= 1e-4f) return -1;
  const int M1 = g1->OH * g1->OW;
  // layer 1: 32 -> 32 VALID on the conv1 output
  if (g2->C != 32 || g2->N != 32 || g2->PT != 0 || g2->PL != 0 || g2->H != g1->OH || g2->W != g1->OW) return -1;
  if ((g2->OH - 1) * g2->S + g2->KH > g2->H || (g2->OW - 1) * g2->S + g2->KW > g2->W) return -1;
  // layer 2: 32 -> 32|64 VALID stride 1 on the conv2 output
  if (g3->C != 32 || (g3->N != 32 && g3->N != 64) || g3->S != 1 || g3->PT != 0 || g3->PL != 0) return -1;
  if (g3->H != g2->OH || g3->W != g2->OW || g3->OH + g3->KH - 1 > g3->H || g3->OW + g3->KW - 1 > g3->W) return -1;
  const int M2 = g2->OH * g2->OW, M3 = g3->OH * g3->OW;
  const int ntiles2 = (M2 + 31) / 32, ntiles3 = ((M3 + 31) / 32) * (g3->N / 32);
  if (2 * ntiles2 > kTW || 2 * ntiles3 > kTW || M1 > 32 * kTW * kTrunkTiles1) return -1;
  if (g2->KH * g2->KW > 64 * kTW || g3->KH * g3->KW > 64 * kTW || g2->KH * g2->KW < 2 || g3->KH * g3->KW < 2) return -1;
  TrunkFwdArgs a;
  a.in = static_cast<const uint8_t*>(in); a.idx = idx;
  a.w1 = w1; a.b1 = b1; a.y1 = y1;
  a.H = g1->H; a.W = g1->W; a.OH1 = g1->OH; a.OW1 = g1->OW; a.S1 = g1->S; a.KH1 = g1->KH; a.act1 = g1->act;
  a.xs = 1.f / xf->std;
  a.w2 = w2; a.b2 = b2; a.y2 = y2;
  a.KH2 = g2->KH; a.KW2 = g2->KW; a.S2 = g2->S; a.OH2 = g2->OH; a.OW2 = g2->OW; a.act2 = g2->act;
  const int PH1 = (g1->OH + g2->S - 1) / g2->S;
  a.PW1 = (g1->OW + g2->S - 1) / g2->S;
  a.PHW1 = PH1 * a.PW1;
  a.CS1 = (g2->S * g2->S * a.PHW1) | 1;             // odd channel stride: conflict-free channel-strided writes
  a.d_ow1 = make_fastdiv(g1->OW); a.d_s2 = make_fastdiv(g2->S); a.d_ow2 = make_fastdiv(g2->OW);
  a.w3 = w3; a.b3 = b3; a.y3 = y3;
  a.KH3 = g3->KH; a.KW3 = g3->KW; a.OH3 = g3->OH; a.OW3 = g3->OW; a.N3 = g3->N; a.act3 = g3->act;
  // act2 rows are read up to (OH3-1+KH3-1)*OW2 + OW3-1+KW3-1 < M2: in range by the VALID checks above
  a.CS2 = M2 | 1;
  a.d_ow3 = make_fastdiv(g3->OW);
  a.B = B;
  // region 0: frame stack, later act2 + conv2 partials; region 1: act1, later conv3 partials; then the tables
  size_t r0 = (size_t)32 * a.CS2 * 4 + (size_t)ntiles2 * 4096;
  if ((size_t)HWC > r0) r0 = HWC;
  r0 = (r0 + 15) & ~(size_t)15;
  size_t r1 = (size_t)32 * a.CS1 * 4;
  if ((size_t)ntiles3 * 4096 > r1) r1 = (size_t)ntiles3 * 4096;
  r1 = (r1 + 15) & ~(size_t)15;
  a.off_act1 = (int)r0;
  a.off_tab = (int)(r0 + r1);
  const size_t ldsb = r0 + r1 + (((size_t)M1 * 2 + 15) & ~(size_t)15) + (size_t)(g2->KH * g2->KW + g3->KH * g3->KW) * 4;
  if (ldsb > 80 * 1024 || M1 * 1 > 65535 || a.CS1 * 32 > 65535 * 4) return -1;
  const bool fixed = (a.CS1 == 401 && a.CS2 == 81);       // PpoCnn 84x84: compile-time channel strides
  static bool attr_done = false;
  if (!attr_done) {
    XT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trunk_fwd_kernel<401, 81>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    XT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trunk_fwd_kernel<0, 0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    attr_done = true;
  }
  if (fixed) hipLaunchKernelGGL((trunk_fwd_kernel<401, 81>), dim3(B), dim3(64 * kTW), ldsb, st, a);
  else hipLaunchKernelGGL((trunk_fwd_kernel<0, 0>), dim3(B), dim3(64 * kTW), ldsb, st, a);
  XT_LAUNCH_CHECK();
  return 0;
}

}  // namespace xt
