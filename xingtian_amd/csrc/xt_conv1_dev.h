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
