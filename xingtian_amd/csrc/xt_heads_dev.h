// Device-side body of the head weight-gradient partial sums, shared by xt_heads.hip (stand-alone launch)
// and xt_igemm.hip (horizontally fused into the last trunk layer's backward launch).
#pragma once
#include "xt_common.h"

namespace xt {

struct HeadWgArgs {
  const float *f_pi, *f_v, *dlogits, *dvalue;
  float *slab_pi, *slab_v;
  long long stride_pi, stride_v;
  int B, F, A, gx, nchunk;      // gx = ceil(F/64) feature blocks, nchunk = ceil(B/8) batch chunks
};

// block (bx, by): 64 features x one chunk of 8 samples; slab_pi[chunk][F*A + A], slab_v[chunk][F + 1]
// are summed later by grads_finish_kernel.  Needs 4*64*9 floats of LDS.
__device__ __forceinline__ void heads_wgrad_partial_body(const HeadWgArgs& h, const int bx, const int by, float* smem) {
  float(*red)[64][9] = reinterpret_cast<float(*)[64][9]>(smem);
  const int fl = threadIdx.x & 63, bg = threadIdx.x >> 6;
  const int F = h.F, A = h.A, B = h.B;
  const int f = bx * 64 + fl;
  const bool fok = f < F;
  const int b0 = by * 8;
  float* spi = h.slab_pi + (size_t)by * h.stride_pi;
  float* svp = h.slab_v + (size_t)by * h.stride_v;
  for (int a0 = 0; a0 <= A; a0 += 8) {
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int b = b0 + bg + 4 * r;
      if (b < B) {
        const float xp = fok ? h.f_pi[(size_t)b * F + f] : 0.f;
        const float xv = fok ? h.f_v[(size_t)b * F + f] : 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int a = a0 + q;
          if (a < A) acc[q] = fmaf(xp, h.dlogits[(size_t)b * A + a], acc[q]);
          else if (a == A) acc[q] = fmaf(xv, h.dvalue[b], acc[q]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) red[bg][fl][q] = acc[q];
    __syncthreads();
    if (bg == 0 && fok) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int a = a0 + q;
        const float s = red[0][fl][q] + red[1][fl][q] + red[2][fl][q] + red[3][fl][q];
        if (a < A) spi[(size_t)f * A + a] = s;
        else if (a == A) svp[f] = s;
      }
    }
    __syncthreads();
  }
  if (bx == 0 && (int)threadIdx.x <= A) {
    const int a = threadIdx.x;
    float s = 0.f;
    for (int r = 0; r < 8; ++r) {
      const int b = b0 + r;
      if (b < B) s += (a < A) ? h.dlogits[(size_t)b * A + a] : h.dvalue[b];
    }
    if (a < A) spi[(size_t)F * A + a] = s; else svp[F] = s;
  }
}

}  // namespace xt
