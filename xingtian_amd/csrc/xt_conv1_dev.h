// bf16x3 helpers of the first-layer (uint8 frame-stack) kernels, shared by xt_conv1.hip and xt_trunk.hip.
#pragma once
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

// 4 consecutive-k fp32 values -> three bf16 planes (x = x1 + x2 + x3 by truncation: 3 x 8 = 24 significant bits),
// 8 bytes each, at p0 + plane * plane_stride.  Every bf16 x bf16 product is exact in fp32, so
// x*w = x1*w1 + (x1*w2 + x2*w1) + (x1*w3 + x2*w2 + x3*w1) + terms below 2^-24 relative ("bf16x6").
__device__ __forceinline__ void split3_store(uint8_t* p0, int plane_stride, const float4 v) {
#if defined(XT_ABL) && XT_ABL == 1     // ablation: no split arithmetic (timing probe, wrong numerics)
  const uint2 w = make_uint2(pack_hi16(v.x, v.y), pack_hi16(v.z, v.w));
  *reinterpret_cast<uint2*>(p0) = w; *reinterpret_cast<uint2*>(p0 + plane_stride) = w; *reinterpret_cast<uint2*>(p0 + 2 * plane_stride) = w;
  return;
#endif
  const float rx = v.x - trunc_bf16(v.x), ry = v.y - trunc_bf16(v.y), rz = v.z - trunc_bf16(v.z), rw = v.w - trunc_bf16(v.w);
  const float qx = rx - trunc_bf16(rx), qy = ry - trunc_bf16(ry), qz = rz - trunc_bf16(rz), qw = rw - trunc_bf16(rw);
  *reinterpret_cast<uint2*>(p0) = make_uint2(pack_hi16(v.x, v.y), pack_hi16(v.z, v.w));
  *reinterpret_cast<uint2*>(p0 + plane_stride) = make_uint2(pack_hi16(rx, ry), pack_hi16(rz, rw));
  *reinterpret_cast<uint2*>(p0 + 2 * plane_stride) = make_uint2(pack_hi16(qx, qy), pack_hi16(qz, qw));
}
// the same split of 8 values held in registers -> three MFMA operands
__device__ __forceinline__ void split3_regs(const float4 lo, const float4 hi, bf16x8 (&out)[3]) {
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  float r[8], q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { r[i] = v[i] - trunc_bf16(v[i]); q[i] = r[i] - trunc_bf16(r[i]); }
  BF8 b0, b1, b2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b0.u[i] = pack_hi16(v[2 * i], v[2 * i + 1]);
    b1.u[i] = pack_hi16(r[2 * i], r[2 * i + 1]);
    b2.u[i] = pack_hi16(q[2 * i], q[2 * i + 1]);
  }
  out[0] = b0.v; out[1] = b1.v; out[2] = b2.v;
}
// six-term accumulate, smallest terms first
__device__ __forceinline__ f32x16 mfma_bf16x6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
  return acc;
}

// global -> LDS copy of one frame stack: 8 x 16-byte loads per thread are issued back to back (one memory
// latency for the whole image instead of one per loop iteration), then written to LDS
template <int NT = 256>
__device__ __forceinline__ void stage_image(const uint4* __restrict__ src, uint4* dst, int n16, int t) {
  constexpr int U = 2048 / NT;           // loads in flight per thread: 2048 x 16 B = 32 KB per pass
  for (int base = 0; base < n16; base += NT * U) {
    uint4 v[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int i = base + t + NT * q;
      v[q] = src[i < n16 ? i : 0];
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int i = base + t + NT * q;
      if (i < n16) dst[i] = v[q];
    }
  }
}

}  // namespace xt
