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

namespace xt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

union BF8 {
  uint32_t u[4];
  bf16x8 v;
};

// two floats -> packed {bf16(lo) in bits 0..15, bf16(hi) in bits 16..31} by truncation (upper halves)
__device__ __forceinline__ uint32_t pack_hi16(float lo, float hi) {
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// 8 consecutive bytes (two dwords) -> 8 bf16 (exact)
__device__ __forceinline__ bf16x8 bytes_to_bf16x8(uint32_t d0, uint32_t d1) {
  BF8 r;
  r.u[0] = pack_hi16((float)(d0 & 0xffu), (float)((d0 >> 8) & 0xffu));
  r.u[1] = pack_hi16((float)((d0 >> 16) & 0xffu), (float)(d0 >> 24));
  r.u[2] = pack_hi16((float)(d1 & 0xffu), (float)((d1 >> 8) & 0xffu));
  r.u[3] = pack_hi16((float)((d1 >> 16) & 0xffu), (float)(d1 >> 24));
  return r.v;
}

struct C1FwdArgs {
  const uint8_t* in;
  const int32_t* idx;
  const float* w;      // [KH*32][32]
  const float* bias;   // [32]
  float* y;            // [B*OH*OW][32]
  int B, H, W, OH, OW, S, KH, act;
  float xs, xb;        // y = act(acc*xs + xb*colsum(W) + bias)
};

constexpr int kC1MaxTiles = 4;   // 32-pixel tiles per wave -> OH*OW <= 4*4*32 = 512

__global__ __launch_bounds__(256, 2) void conv_u8c4k8_fwd_bf16x3_kernel(const C1FwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t limg[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b = blockIdx.x;
  const int HWC = p.H * p.W * 4, Wrow = p.W * 4, OHOW = p.OH * p.OW;
  // ---- stage the frame stack (coalesced 16-byte loads, minibatch gather fused)
  {
    const size_t s = p.idx ? (size_t)p.idx[b] : (size_t)b;
    const uint4* src = reinterpret_cast<const uint4*>(p.in + s * (size_t)HWC);
    uint4* dst = reinterpret_cast<uint4*>(limg);
    const int n16 = HWC >> 4;
    for (int i = t; i < n16; i += 256) dst[i] = src[i];
  }
  const int ntiles = (OHOW + 31) >> 5;
  const int il = lane & 31, h = lane >> 5;
  // per-tile byte offset of this lane's pixel (top-left of its receptive field)
  int poff[kC1MaxTiles];
#pragma unroll
  for (int ti = 0; ti < kC1MaxTiles; ++ti) {
    const int pix = (wave + 4 * ti) * 32 + il;
    const int pp = pix < OHOW ? pix : 0;
    const int oy = pp / p.OW, ox = pp - oy * p.OW;
    poff[ti] = (p.S * oy * p.W + p.S * ox) * 4 + 8 * h;
  }
  f32x16 acc[kC1MaxTiles];
#pragma unroll
  for (int ti = 0; ti < kC1MaxTiles; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;

  const int nsteps = 2 * p.KH;              // 16 reduction elements per step = half a kernel row
  const float* wl = p.w + (size_t)(8 * h) * 32 + il;   // this lane's column n = il, rows 8h + j
  float wcur[8], wnext[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) wcur[j] = wl[j * 32];
  float sumw = 0.f;
  __syncthreads();

  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) {
#pragma unroll
      for (int j = 0; j < 8; ++j) wnext[j] = wl[(size_t)((s + 1) * 16 + j) * 32];
    }
    // exact 3-way bf16 split of the 8 weights of this lane
    BF8 b1, b2, b3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float w0 = wcur[2 * q], w1 = wcur[2 * q + 1];
      sumw += w0 + w1;
      const float r0 = w0 - trunc_bf16(w0), r1 = w1 - trunc_bf16(w1);
      const float q0 = r0 - trunc_bf16(r0), q1 = r1 - trunc_bf16(r1);
      b1.u[q] = pack_hi16(w0, w1);
      b2.u[q] = pack_hi16(r0, r1);
      b3.u[q] = pack_hi16(q0, q1);
    }
    const int koff = (s >> 1) * Wrow + (s & 1) * 16;
#pragma unroll
    for (int ti = 0; ti < kC1MaxTiles; ++ti) {
      if (wave + 4 * ti < ntiles) {     // wave-uniform
        const uint2 d = *reinterpret_cast<const uint2*>(limg + poff[ti] + koff);
        const bf16x8 a = bytes_to_bf16x8(d.x, d.y);
        acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1.v, acc[ti], 0, 0, 0);
        acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2.v, acc[ti], 0, 0, 0);
        acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b3.v, acc[ti], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) wcur[j] = wnext[j];
  }

  // ---- epilogue: input transform on the accumulator, bias, activation
  const float colsum = sumw + __shfl_xor(sumw, 32, 64);
  const float cb = fmaf(p.xb, colsum, p.bias[il]);
#pragma unroll
  for (int ti = 0; ti < kC1MaxTiles; ++ti) {
    if (wave + 4 * ti < ntiles) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pix = (wave + 4 * ti) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (pix < OHOW)
          p.y[((size_t)b * OHOW + pix) * 32 + il] = act_apply(fmaf(acc[ti][r], p.xs, cb), p.act);
      }
    }
  }
}

// returns 0 launched, 1 error, -1 geometry not handled by this kernel
int launch_conv1_fwd_bf16x3(const xt_conv_geom* g, const xt_input_xform* xf, int B, const void* in,
                            const int32_t* idx, const float* w, const float* bias, float* y, hipStream_t st) {
  if (!xf || !xf->is_u8 || g->C != 4 || g->KW != 8 || g->N != 32 || g->PT != 0 || g->PL != 0) return -1;
  if ((g->OH - 1) * g->S + g->KH > g->H || (g->OW - 1) * g->S + g->KW > g->W) return -1;
  const int HWC = g->H * g->W * 4;
  if (HWC % 16 != 0 || HWC > 64 * 1024 || (g->W * 4) % 8 != 0 || (g->S * 4) % 8 != 0) return -1;
  if (g->OH * g->OW > 32 * 4 * kC1MaxTiles) return -1;
  C1FwdArgs a;
  a.in = static_cast<const uint8_t*>(in); a.idx = idx; a.w = w; a.bias = bias; a.y = y;
  a.B = B; a.H = g->H; a.W = g->W; a.OH = g->OH; a.OW = g->OW; a.S = g->S; a.KH = g->KH; a.act = g->act;
  const float mean = fabsf(xf->mean) >= 1e-4f ? xf->mean : 0.f;
  a.xs = 1.f / xf->std; a.xb = -mean * a.xs;
  hipLaunchKernelGGL(conv_u8c4k8_fwd_bf16x3_kernel, dim3(B), dim3(256), HWC, st, a);
  XT_LAUNCH_CHECK();
  return 0;
}

int launch_conv1_wgrad_bf16x3(const xt_conv_geom*, const xt_input_xform*, int, const void*, const int32_t*,
                              const float*, float*, float*, int, int*, hipStream_t) {
  return -1;   // not implemented yet
}

}  // namespace xt
