// Geometry shared by the implicit-GEMM kernel families (xt_igemm.hip: LDS-tiled, xt_direct.hip: register-direct).
#pragma once
#include <math.h>
#include "xt_common.h"

namespace xt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Geom {
  int B, H, W, C, KH, KW, S, PT, PL, OH, OW, N, act;
  int K, KWC, M, OHOW;
  FastDiv d_ohow, d_ow, d_kwc, d_c, d_n;
  float xs, xb;          // uint8 -> f32 transform: x*xs + xb  (xs = 1/std, xb = -mean/std)
  int HWC;
};

static inline int make_geom(const xt_conv_geom* g, const xt_input_xform* xf, int B, Geom* o) {
  XT_REQUIRE(g && B > 0, "igemm: bad geometry/batch");
  o->B = B; o->H = g->H; o->W = g->W; o->C = g->C; o->KH = g->KH; o->KW = g->KW; o->S = g->S;
  o->PT = g->PT; o->PL = g->PL; o->OH = g->OH; o->OW = g->OW; o->N = g->N; o->act = g->act;
  o->K = g->KH * g->KW * g->C; o->KWC = g->KW * g->C; o->OHOW = g->OH * g->OW;
  XT_REQUIRE(g->C % 4 == 0, "igemm: input channels C=%d must be a multiple of 4", g->C);
  XT_REQUIRE(g->N % 4 == 0, "igemm: output channels N=%d must be a multiple of 4", g->N);
  XT_REQUIRE(g->S >= 1 && g->KH >= 1 && g->KW >= 1, "igemm: bad kernel/stride");
  long long m = (long long)B * o->OHOW;
  XT_REQUIRE(m * (long long)o->OHOW < (1ll << 32) && m < (1ll << 30), "igemm: M=%lld too large", m);
  XT_REQUIRE((long long)B * g->H * g->W * g->C < (1ll << 31), "igemm: activation tensor too large");
  o->M = (int)m;
  o->d_ohow = make_fastdiv(o->OHOW); o->d_ow = make_fastdiv(o->OW);
  o->d_kwc = make_fastdiv(o->KWC); o->d_c = make_fastdiv(o->C); o->d_n = make_fastdiv(o->N);
  const float mean = (xf && fabsf(xf->mean) >= 1e-4f) ? xf->mean : 0.f;   // state_transform: |mean|<1e-4 -> x/std
  o->xs = xf ? 1.f / xf->std : 1.f;
  o->xb = -mean * o->xs;
  o->HWC = g->H * g->W * g->C;
  // BYTES, not elements: the write-through stores (store4_wt, xt_common.h) address the output through a raw buffer
  // descriptor with a 32-bit byte offset, i.e. 2 GiB (an 8 GiB tensor passed the old element bound and the
  // hardware silently dropped the out-of-range stores)
  XT_REQUIRE(m * (long long)g->N * 4 < (1ll << 31), "igemm: output tensors of 2 GiB or more are not supported (batch %d)", B);
  return 0;
}

// does the receptive field ever leave the image (TF SAME padding)?
static inline bool is_padded(const Geom& g) {
  return g.PT > 0 || g.PL > 0 || (g.OH - 1) * g.S - g.PT + g.KH > g.H || (g.OW - 1) * g.S - g.PL + g.KW > g.W;
}

}  // namespace xt
