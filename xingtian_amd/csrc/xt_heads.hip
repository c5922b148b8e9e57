// Policy/value heads, PPO and IMPALA(v-trace) losses and their gradients, GAE.
// All HBM/latency-bound; wave64 shuffles for the reductions, no atomics (deterministic).
#include "xt_common.h"
#include "xt_heads_dev.h"

namespace xt {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------- heads forward
// one wave per sample; lanes stride the feature axis
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ f_pi, const float* __restrict__ f_v,
                                                        int B, int F, int A, const float* __restrict__ wpi,
                                                        const float* __restrict__ bpi, const float* __restrict__ wv,
                                                        const float* __restrict__ bv, float* __restrict__ logits,
                                                        float* __restrict__ value) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* fp = f_pi + (size_t)b * F;
  const float* fv = f_v + (size_t)b * F;
  for (int a = 0; a < A; ++a) {
    float s = 0.f;
    for (int f = lane; f < F; f += 64) s = fmaf(fp[f], wpi[(size_t)f * A + a], s);
    s = wave_sum(s);
    if (lane == 0) logits[(size_t)b * A + a] = s + bpi[a];
  }
  float s = 0.f;
  for (int f = lane; f < F; f += 64) s = fmaf(fv[f], wv[f], s);
  s = wave_sum(s);
  if (lane == 0) value[b] = s + bv[0];
}

// ---------------------------------------------------------------- PPO loss
// one thread per sample (A is tiny); tf_dist.py:103-113 + model/ppo/__init__.py:4-25
__global__ __launch_bounds__(256) void ppo_loss_kernel(const float* __restrict__ logits, const float* __restrict__ value,
                                                       int B, int A, const int32_t* __restrict__ idx,
                                                       const int32_t* __restrict__ action, const float* __restrict__ old_logp,
                                                       const double* __restrict__ adv, const float* __restrict__ old_v,
                                                       const double* __restrict__ target_v, float clip_ratio,
                                                       float ent_coef, float vf_clip, float critic_coef, float inv_b,
                                                       float* __restrict__ dlogits, float* __restrict__ dvalue,
                                                       float* __restrict__ terms) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const int s = idx ? idx[b] : b;
  const float* lg = logits + (size_t)b * A;
  float mx = lg[0];
  for (int a = 1; a < A; ++a) mx = fmaxf(mx, lg[a]);
  float z = 0.f;
  for (int a = 0; a < A; ++a) z += expf(lg[a] - mx);
  const float logz = logf(z);
  const int act = action[s];
  float ent = 0.f;
  for (int a = 0; a < A; ++a) {
    const float rl = lg[a] - mx;
    ent += (expf(rl) / z) * (logz - rl);
  }
  const float logp = (lg[act] - mx) - logz;
  const float advf = (float)adv[s];            // float32 placeholder cast (ppo.py:65-68)
  const float tv = (float)target_v[s];
  const float ov = old_v[s];
  const float ratio = expf(logp - old_logp[s]);
  const float surr1 = ratio * advf;
  const float rc = fminf(fmaxf(ratio, 1.f - clip_ratio), 1.f + clip_ratio);
  const float surr2 = rc * advf;
  const float surr = fminf(surr1, surr2);
  const bool first = surr1 <= surr2;
  const bool in_rng = (ratio >= 1.f - clip_ratio) && (ratio <= 1.f + clip_ratio);
  const float dsurr = (first || in_rng) ? advf : 0.f;
  const float dlogp = -(dsurr * ratio) * inv_b;
  const float v = value[b];
  const float d1 = v - tv;
  const float vf1 = d1 * d1;
  const float vcl = ov + fminf(fmaxf(v - ov, -vf_clip), vf_clip);
  const float d2 = vcl - tv;
  const float vf2 = d2 * d2;
  const bool take1 = vf1 >= vf2;
  const bool in_v = fabsf(v - ov) <= vf_clip;
  const float dv = take1 ? 2.f * d1 : (in_v ? 2.f * d2 : 0.f);
  dvalue[b] = critic_coef * 0.5f * inv_b * dv;
  for (int a = 0; a < A; ++a) {
    const float rl = lg[a] - mx;
    const float pa = expf(rl) / z;
    const float lpa = rl - logz;
    const float onehot = (a == act) ? 1.f : 0.f;
    // d(-ent_coef*mean(H))/dlogit = +ent_coef/B * p*(log p + H)
    dlogits[(size_t)b * A + a] = dlogp * (onehot - pa) + ent_coef * inv_b * (pa * (lpa + ent));
  }
  float* tm = terms + (size_t)b * 4;
  tm[0] = surr; tm[1] = ent; tm[2] = fmaxf(vf1, vf2); tm[3] = 0.f;
}

// The same loss over DiagGaussianDist (xt/model/tf_dist.py:47-87; dist_param = concat([pi_latent,
// pi_latent*0 + pi_logstd]), xt/model/ppo/ppo.py:75-79).  One thread per sample.  dls_rows [B][ldls] gets every
// sample's share of d loss / d pi_logstd (surrogate part + the -ent_coef * mean(entropy) part); the gradient is
// their fixed-order sum (grads_finish_kernel treats the rows as B partial slabs).
__global__ __launch_bounds__(256) void ppo_loss_gauss_kernel(const float* __restrict__ mean, const float* __restrict__ log_std,
                                                             const float* __restrict__ value, int B, int A,
                                                             const int32_t* __restrict__ idx,
                                                             const float* __restrict__ action,
                                                             const float* __restrict__ old_logp,
                                                             const double* __restrict__ adv, const float* __restrict__ old_v,
                                                             const double* __restrict__ target_v, float clip_ratio,
                                                             float ent_coef, float vf_clip, float critic_coef, float inv_b,
                                                             float* __restrict__ dmean, float* __restrict__ dvalue,
                                                             float* __restrict__ dls_rows, int ldls,
                                                             float* __restrict__ terms) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const int s = idx ? idx[b] : b;
  const float* mu = mean + (size_t)b * A;
  const float* x = action + (size_t)s * A;
  // neglog_prob = 0.5*log(2*pi)*A + 0.5*sum(((x-mean)/std)^2) + sum(log_std)        (tf_dist.py:66-69)
  float ssq = 0.f, sls = 0.f, ent = 0.f;
  for (int a = 0; a < A; ++a) {
    const float ls = log_std[a];
    const float z = (x[a] - mu[a]) / expf(ls);
    ssq += z * z;
    sls += ls;
    ent += ls + 1.4189385332046727f;               // 0.5*(log(2*pi)+1)                (tf_dist.py:74-75)
  }
  const float logp = -(0.9189385332046727f * (float)A + 0.5f * ssq + sls);
  const float advf = (float)adv[s];
  const float tv = (float)target_v[s];
  const float ov = old_v[s];
  const float ratio = expf(logp - old_logp[s]);
  const float surr1 = ratio * advf;
  const float rc = fminf(fmaxf(ratio, 1.f - clip_ratio), 1.f + clip_ratio);
  const float surr2 = rc * advf;
  const float surr = fminf(surr1, surr2);
  const bool first = surr1 <= surr2;
  const bool in_rng = (ratio >= 1.f - clip_ratio) && (ratio <= 1.f + clip_ratio);
  const float dsurr = (first || in_rng) ? advf : 0.f;
  const float dlogp = -(dsurr * ratio) * inv_b;
  const float v = value[b];
  const float d1 = v - tv;
  const float vf1 = d1 * d1;
  const float vcl = ov + fminf(fmaxf(v - ov, -vf_clip), vf_clip);
  const float d2 = vcl - tv;
  const float vf2 = d2 * d2;
  const bool take1 = vf1 >= vf2;
  const bool in_v = fabsf(v - ov) <= vf_clip;
  const float dv = take1 ? 2.f * d1 : (in_v ? 2.f * d2 : 0.f);
  dvalue[b] = critic_coef * 0.5f * inv_b * dv;
  for (int a = 0; a < A; ++a) {
    const float sd = expf(log_std[a]);
    const float z = (x[a] - mu[a]) / sd;
    dmean[(size_t)b * A + a] = dlogp * z / sd;                                   // d logp / d mean = (x-mean)/std^2
    dls_rows[(size_t)b * ldls + a] = dlogp * (z * z - 1.f) - ent_coef * inv_b;   // d logp / d log_std = z^2 - 1; dH/dls = 1
  }
  float* tm = terms + (size_t)b * 4;
  tm[0] = surr; tm[1] = ent; tm[2] = fmaxf(vf1, vf2); tm[3] = 0.f;
}

// single block, fixed-order tree: loss scalars from the per-sample terms
__global__ __launch_bounds__(256) void ppo_loss_reduce_kernel(const float* __restrict__ terms, int B, float ent_coef,
                                                              float critic_coef, float inv_b, float* __restrict__ out,
                                                              float* __restrict__ acc) {
  __shared__ float sh[3][256];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    s0 += terms[(size_t)b * 4 + 0]; s1 += terms[(size_t)b * 4 + 1]; s2 += terms[(size_t)b * 4 + 2];
  }
  sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1; sh[2][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
      sh[2][threadIdx.x] += sh[2][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float surr = sh[0][0] * inv_b, ent = sh[1][0] * inv_b, vf = 0.5f * sh[2][0] * inv_b;
    const float actor = -surr - ent_coef * ent;
    const float loss = actor + critic_coef * vf;
    out[0] = loss; out[1] = actor; out[2] = vf; out[3] = ent;
    if (acc) { acc[0] += loss; acc[1] += 1.f; }
  }
}

// ---------------------------------------------------------------- IMPALA v-trace loss
// one block per trajectory, thread t = time step.  Phases: (1) per-step rho/c/delta in
// parallel, (2) serial reverse scan by thread 0 over LDS, (3) per-step pg_adv + grads.
template <int MAXT>
__global__ __launch_bounds__(MAXT) void impala_loss_kernel(const float* __restrict__ logits, const float* __restrict__ baseline,
                                                           const float* __restrict__ bp_logits, const int32_t* __restrict__ action,
                                                           const uint8_t* __restrict__ done, const float* __restrict__ reward,
                                                           int T, int A, float gamma, float* __restrict__ dlogits,
                                                           float* __restrict__ dbaseline, float* __restrict__ traj_loss,
                                                           float* __restrict__ vs_out, float* __restrict__ pg_out) {
  __shared__ float s_delta[MAXT], s_dc[MAXT], s_vs[MAXT + 1], s_red[MAXT];
  const int t = threadIdx.x;
  const int traj = blockIdx.x;
  const int Tm = T - 1;                      // steps that carry loss; step T-1 is the bootstrap only
  const size_t base = (size_t)traj * T;
  float rho = 0.f, disc = 0.f, rew = 0.f, val = 0.f, ce = 0.f, ent = 0.f, logz = 0.f, mx = 0.f, z = 1.f;
  int act = 0;
  if (t < Tm) {
    const float* lg = logits + (base + t) * A;
    const float* bl = bp_logits + (base + t) * A;
    act = action[base + t];
    mx = lg[0];
    float bmx = bl[0];
    for (int a = 1; a < A; ++a) { mx = fmaxf(mx, lg[a]); bmx = fmaxf(bmx, bl[a]); }
    z = 0.f;
    float bz = 0.f;
    for (int a = 0; a < A; ++a) { z += expf(lg[a] - mx); bz += expf(bl[a] - bmx); }
    logz = logf(z);
    const float tlp = (lg[act] - mx) - logz;
    const float blp = (bl[act] - bmx) - logf(bz);
    ce = -tlp;
    rho = expf(tlp - blp);
    disc = done[base + t] ? 0.f : gamma;
    rew = fminf(fmaxf(reward[base + t], -1.f), 1.f);
    val = baseline[base + t];
    const float nval = baseline[base + t + 1];          // t+1 == T-1 -> bootstrap value
    const float crho = fminf(1.f, rho);
    s_delta[t] = crho * (rew + disc * nval - val);
    s_dc[t] = disc * fminf(1.f, rho);
    for (int a = 0; a < A; ++a) {
      const float rl = lg[a] - mx;
      ent += (expf(rl) / z) * (logz - rl);
    }
  }
  __syncthreads();
  if (t == 0) {
    float acc = 0.f;
    for (int q = Tm - 1; q >= 0; --q) {
      acc = s_delta[q] + s_dc[q] * acc;
      s_vs[q] = acc + baseline[base + q];
    }
    s_vs[Tm] = baseline[base + Tm];                     // bootstrap
  }
  __syncthreads();
  float lterm = 0.f;
  if (t < Tm) {
    const float vs = s_vs[t], vsn = s_vs[t + 1];
    const float pg = fminf(1.f, rho) * (rew + disc * vsn - val);
    const float* lg = logits + (base + t) * A;
    for (int a = 0; a < A; ++a) {
      const float rl = lg[a] - mx;
      const float pa = expf(rl) / z;
      const float lpa = rl - logz;
      const float onehot = (a == act) ? 1.f : 0.f;
      dlogits[(base + t) * A + a] = pg * (pa - onehot) + 0.01f * (pa * (lpa + ent));
    }
    dbaseline[base + t] = 0.5f * (val - vs);
    const float dvv = vs - val;
    lterm = ce * pg + 0.5f * (0.5f * dvv * dvv) + 0.01f * (-ent);
    if (vs_out) vs_out[(size_t)traj * Tm + t] = vs;
    if (pg_out) pg_out[(size_t)traj * Tm + t] = pg;
  } else if (t == Tm) {
    for (int a = 0; a < A; ++a) dlogits[(base + t) * A + a] = 0.f;
    dbaseline[base + t] = 0.f;
  }
  s_red[t] = lterm;
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
    for (int q = 0; q < T; ++q) s += s_red[q];
    traj_loss[traj] = s;
  }
}

__global__ void impala_loss_reduce_kernel(const float* __restrict__ traj_loss, int n, float* __restrict__ out,
                                          float* __restrict__ acc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += traj_loss[i];
    out[0] = s;
    if (acc) { acc[0] += s; acc[1] += 1.f; }
  }
}

// ---------------------------------------------------------------- heads backward
// gradient w.r.t. trunk features, times the producer's activation gradient
__global__ __launch_bounds__(256) void heads_dfeat_kernel(const float* __restrict__ f_pi, const float* __restrict__ f_v,
                                                          int B, int F, int A, const float* __restrict__ wpi,
                                                          const float* __restrict__ wv, const float* __restrict__ dlogits,
                                                          const float* __restrict__ dvalue, int act_prev, int shared,
                                                          float* __restrict__ df_pi, float* __restrict__ df_v) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * F) return;
  const int b = e / F, f = e - b * F;
  float s = 0.f;
  for (int a = 0; a < A; ++a) s = fmaf(dlogits[(size_t)b * A + a], wpi[(size_t)f * A + a], s);
  const float sv = dvalue[b] * wv[f];
  if (shared) {
    df_pi[e] = (s + sv) * act_grad(f_pi[e], act_prev);
  } else {
    df_pi[e] = s * act_grad(f_pi[e], act_prev);
    df_v[e] = sv * act_grad(f_v[e], act_prev);
  }
}

// head weight gradients: block = 64 features x 4 batch groups; outputs in chunks of 8
__global__ __launch_bounds__(256) void heads_wgrad_kernel(const float* __restrict__ f_pi, const float* __restrict__ f_v,
                                                          int B, int F, int A, const float* __restrict__ dlogits,
                                                          const float* __restrict__ dvalue, float* __restrict__ dwpi,
                                                          float* __restrict__ dbpi, float* __restrict__ dwv,
                                                          float* __restrict__ dbv) {
  __shared__ float red[4][64][9];
  const int fl = threadIdx.x & 63, bg = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + fl;
  const bool fok = f < F;
  // outputs 0..A-1: pi columns, output A: value column
  for (int a0 = 0; a0 <= A; a0 += 8) {
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    for (int b = bg; b < B; b += 4) {
      const float xp = fok ? f_pi[(size_t)b * F + f] : 0.f;
      const float xv = fok ? f_v[(size_t)b * F + f] : 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int a = a0 + q;
        if (a < A) acc[q] = fmaf(xp, dlogits[(size_t)b * A + a], acc[q]);
        else if (a == A) acc[q] = fmaf(xv, dvalue[b], acc[q]);
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
        if (a < A) dwpi[(size_t)f * A + a] = s;
        else if (a == A) dwv[f] = s;
      }
    }
    __syncthreads();
  }
  // bias grads: block 0, wave-parallel over batch
  if (blockIdx.x == 0) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int a = w; a <= A; a += 4) {
      float s = 0.f;
      for (int b = lane; b < B; b += 64) s += (a < A) ? dlogits[(size_t)b * A + a] : dvalue[b];
      s = wave_sum(s);
      if (lane == 0) { if (a < A) dbpi[a] = s; else dbv[0] = s; }
    }
  }
}

// ---------------------------------------------------------------- fused PPO head (one launch per SGD step)
// heads forward + PPO loss + d(logits,value) + gradient w.r.t. the trunk features, one wave per
// sample, lane a <-> action a (A <= 64).  Same arithmetic as heads_fwd_kernel / ppo_loss_kernel /
// heads_dfeat_kernel; the loss scalar is reduced later (norm_finalize_kernel) from `terms`.
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// One wave (= one 64-thread workgroup) per sample, specialised on NQ = ceil(F/64) features per lane, on whether
// the split-K partial slabs of the last trunk layer still have to be summed (PART) and on shared/separate trunks.
// EVERY global load of the wave is issued before the first dependent instruction (the kernel is a pure latency
// chain: 73 % of its wave cycles were s_waitcnt with the loads interleaved with their uses), the only two-hop
// chain being idx -> labels.
constexpr int kHeadMaxA = 8;          // actions handled by the fused kernel (A <= 8)
constexpr int kMaxHeadSplit = 16;     // split-K partial slabs the fused head kernel can finish

template <int NQ, bool PART, bool SHARED>
__global__ __launch_bounds__(64) void ppo_heads_fused_kernel(const PpoHeadArgs p) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int F = p.F, A = p.A;
  const size_t row = (size_t)b * F;
  // ---------------- issue phase (no dependent use in here)
  const int s = p.idx ? p.idx[b] : b;
  float praw[PART ? NQ : 1][kMaxHeadSplit], vraw[(PART && !SHARED) ? NQ : 1][kMaxHeadSplit];
  float xin[NQ], xvin[NQ], tb[NQ], tbv[NQ], wvv[NQ], wp[NQ][kHeadMaxA];
  int fcl[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int f = lane + 64 * q;
    fcl[q] = f < F ? f : 0;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (PART) {
#pragma unroll
      for (int z = 0; z < kMaxHeadSplit; ++z)
        praw[q][z] = p.part_pi[(size_t)(z < p.ksplit_pi ? z : p.ksplit_pi - 1) * p.part_stride + row + fcl[q]];
      tb[q] = p.tbias_pi[fcl[q]];
      if (!SHARED) {
#pragma unroll
        for (int z = 0; z < kMaxHeadSplit; ++z)
          vraw[q][z] = p.part_v[(size_t)(z < p.ksplit_v ? z : p.ksplit_v - 1) * p.part_stride + row + fcl[q]];
        tbv[q] = p.tbias_v[fcl[q]];
      }
    } else {
      xin[q] = p.f_pi[row + fcl[q]];
      if (!SHARED) xvin[q] = p.f_v[row + fcl[q]];
    }
    wvv[q] = p.wv[fcl[q]];
#pragma unroll
    for (int a = 0; a < kHeadMaxA; ++a) wp[q][a] = p.wpi[(size_t)fcl[q] * A + (a < A ? a : 0)];
  }
  const float mybias = p.bpi[lane < A ? lane : 0];
  const float bvv = p.bv[0];
  // labels (second hop of idx)
  const int act = p.action[s];
  const float advf = (float)p.adv[s];
  const float tv = (float)p.target_v[s];
  const float ov = p.old_v[s];
  const float olp = p.old_logp[s];

  // ---------------- features
  float fpi[NQ], fvv[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const bool ok = lane + 64 * q < F;
    float x, xv;
    if (PART) {
      float sacc = 0.f;
#pragma unroll
      for (int z = 0; z < kMaxHeadSplit; z += 4)
        sacc += ((z < p.ksplit_pi ? praw[q][z] : 0.f) + (z + 1 < p.ksplit_pi ? praw[q][z + 1] : 0.f)) +
                ((z + 2 < p.ksplit_pi ? praw[q][z + 2] : 0.f) + (z + 3 < p.ksplit_pi ? praw[q][z + 3] : 0.f));
      x = act_apply(sacc + tb[q], p.act_feat);
      xv = x;
      if (!SHARED) {
        float vacc = 0.f;
#pragma unroll
        for (int z = 0; z < kMaxHeadSplit; z += 4)
          vacc += ((z < p.ksplit_v ? vraw[q][z] : 0.f) + (z + 1 < p.ksplit_v ? vraw[q][z + 1] : 0.f)) +
                  ((z + 2 < p.ksplit_v ? vraw[q][z + 2] : 0.f) + (z + 3 < p.ksplit_v ? vraw[q][z + 3] : 0.f));
        xv = act_apply(vacc + tbv[q], p.act_feat);
      }
    } else {
      x = xin[q];
      xv = SHARED ? x : xvin[q];
    }
    fpi[q] = ok ? x : 0.f;
    fvv[q] = ok ? xv : 0.f;
  }
  // ---------------- logits (lane a keeps logit a) and value
  float acc[kHeadMaxA];
#pragma unroll
  for (int a = 0; a < kHeadMaxA; ++a) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) t = fmaf(fpi[q], wp[q][a], t);
    acc[a] = t;
  }
  float sv = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) sv = fmaf(fvv[q], wvv[q], sv);
  float mylogit = -INFINITY;
#pragma unroll
  for (int a = 0; a < kHeadMaxA; ++a) {
    const float t = wave_sum(acc[a]);
    if (a < A && lane == a) mylogit = t + mybias;
  }
  const float v = wave_sum(sv) + bvv;

  const bool la = lane < A;
  const float mx = wave_max(mylogit);
  const float rl = la ? mylogit - mx : 0.f;
  const float e = la ? expf(rl) : 0.f;
  const float z = wave_sum(e);
  const float logz = logf(z);
  const float pa = e / z;
  const float ent = wave_sum(la ? pa * (logz - rl) : 0.f);
  const float lpa = rl - logz;
  const float logp = __shfl(lpa, act, 64);
  const float ratio = expf(logp - olp);
  const float surr1 = ratio * advf;
  const float rc = fminf(fmaxf(ratio, 1.f - p.clip_ratio), 1.f + p.clip_ratio);
  const float surr2 = rc * advf;
  const bool first = surr1 <= surr2;
  const bool in_rng = (ratio >= 1.f - p.clip_ratio) && (ratio <= 1.f + p.clip_ratio);
  const float dsurr = (first || in_rng) ? advf : 0.f;
  const float dlogp = -(dsurr * ratio) * p.inv_b;
  const float d1 = v - tv, vf1 = d1 * d1;
  const float vcl = ov + fminf(fmaxf(v - ov, -p.vf_clip), p.vf_clip);
  const float d2 = vcl - tv, vf2 = d2 * d2;
  const bool take1 = vf1 >= vf2;
  const bool in_v = fabsf(v - ov) <= p.vf_clip;
  const float dvr = take1 ? 2.f * d1 : (in_v ? 2.f * d2 : 0.f);
  const float dv = p.critic_coef * 0.5f * p.inv_b * dvr;
  const float onehot = (lane == act) ? 1.f : 0.f;
  const float dl = la ? dlogp * (onehot - pa) + p.ent_coef * p.inv_b * (pa * (lpa + ent)) : 0.f;
  // ---------------- d(features) = dlogits . Wpi^T (+ dvalue . Wv^T), times the producer's activation gradient
  float dacc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) dacc[q] = 0.f;
#pragma unroll
  for (int a = 0; a < kHeadMaxA; ++a) {
    const float dla = __shfl(dl, a, 64);          // dl == 0 for lanes >= A
#pragma unroll
    for (int q = 0; q < NQ; ++q) dacc[q] = fmaf(dla, wp[q][a], dacc[q]);
  }
  // ---------------- stores
  if (la) {
    p.logits[(size_t)b * A + lane] = mylogit;
    p.dlogits[(size_t)b * A + lane] = dl;
  }
  if (lane == 0) {
    p.value[b] = v;
    p.dvalue[b] = dv;
    float* tm = p.terms + (size_t)b * 4;
    tm[0] = fminf(surr1, surr2); tm[1] = ent; tm[2] = fmaxf(vf1, vf2); tm[3] = 0.f;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int f = lane + 64 * q;
    if (f < F) {
      if (PART) {
        p.feat_pi_w[row + f] = fpi[q];
        if (!SHARED) p.feat_v_w[row + f] = fvv[q];
      }
      const float svv = dv * wvv[q];
      if (SHARED) {
        p.df_pi[row + f] = (dacc[q] + svv) * act_grad(fpi[q], p.act_prev);
      } else {
        p.df_pi[row + f] = dacc[q] * act_grad(fpi[q], p.act_prev);
        p.df_v[row + f] = svv * act_grad(fvv[q], p.act_prev);
      }
    }
  }
}

// head weight-gradient partial slabs: grid (ceil(F/64), nchunk); body in xt_heads_dev.h
__global__ __launch_bounds__(256) void heads_wgrad_partial_kernel(const HeadWgArgs h) {
  __shared__ float smem[4 * 64 * 9];
  heads_wgrad_partial_body(h, blockIdx.x, blockIdx.y, smem);
}

// returns -1 when the geometry is outside the fused kernel's envelope (caller falls back to the 3 plain kernels)
int launch_ppo_heads_fused(const PpoHeadArgs& a, hipStream_t st) {
  if (a.A > kHeadMaxA || a.F > 512) return -1;
  const bool part = a.part_pi != nullptr;
  const bool shared = a.shared != 0;
  if (part && (a.ksplit_pi > kMaxHeadSplit || (!shared && (!a.part_v || a.ksplit_v > kMaxHeadSplit)))) return -1;
  const int nq = (a.F + 63) / 64;
  const dim3 grid(a.B), blk(64);
#define XT_HEAD(NQV)                                                                                         \
  do {                                                                                                       \
    if (part && shared) hipLaunchKernelGGL((ppo_heads_fused_kernel<NQV, true, true>), grid, blk, 0, st, a);  \
    else if (part) hipLaunchKernelGGL((ppo_heads_fused_kernel<NQV, true, false>), grid, blk, 0, st, a);      \
    else if (shared) hipLaunchKernelGGL((ppo_heads_fused_kernel<NQV, false, true>), grid, blk, 0, st, a);    \
    else hipLaunchKernelGGL((ppo_heads_fused_kernel<NQV, false, false>), grid, blk, 0, st, a);               \
  } while (0)
  if (nq <= 1) XT_HEAD(1); else if (nq <= 2) XT_HEAD(2); else if (nq <= 4) XT_HEAD(4); else XT_HEAD(8);
#undef XT_HEAD
  XT_LAUNCH_CHECK();
  return 0;
}

int launch_heads_dfeat(const float* f_pi, const float* f_v, int B, int F, int A, const float* wpi, const float* wv,
                       const float* dlogits, const float* dvalue, int act_prev, float* df_pi, float* df_v,
                       hipStream_t st) {
  const int shared = (f_pi == f_v) ? 1 : 0;
  hipLaunchKernelGGL(heads_dfeat_kernel, dim3((B * F + 255) / 256), dim3(256), 0, st, f_pi, f_v, B, F, A, wpi, wv,
                     dlogits, dvalue, act_prev, shared, df_pi, df_v);
  XT_LAUNCH_CHECK();
  return 0;
}

int launch_heads_wgrad_partial(const float* f_pi, const float* f_v, int B, int F, int A, const float* dlogits,
                               const float* dvalue, float* slab_pi, long long stride_pi, float* slab_v,
                               long long stride_v, int* nchunk_out, hipStream_t st) {
  const int nchunk = (B + 7) / 8;
  HeadWgArgs h;
  h.f_pi = f_pi; h.f_v = f_v; h.dlogits = dlogits; h.dvalue = dvalue; h.slab_pi = slab_pi; h.slab_v = slab_v;
  h.stride_pi = stride_pi; h.stride_v = stride_v; h.B = B; h.F = F; h.A = A; h.gx = (F + 63) / 64; h.nchunk = nchunk;
  hipLaunchKernelGGL(heads_wgrad_partial_kernel, dim3(h.gx, nchunk), dim3(256), 0, st, h);
  XT_LAUNCH_CHECK();
  *nchunk_out = nchunk;
  return 0;
}

// ---------------------------------------------------------------- fused IMPALA heads (ImpalaCnnOpt: one shared trunk)
// K1: split-K finish of the 11x11 conv (a dense 3872->256) + 1x1-conv policy + dense baseline, one wave per frame --
// the first half of ppo_heads_fused_kernel (same issue-everything-first structure, same arithmetic as
// splitk_finish + heads_fwd_kernel).  impala_cnn_opt.py:131-152.
template <int NQ, bool PART>
__global__ __launch_bounds__(64) void impala_heads_fwd_kernel(const ImpalaHeadArgs p) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int F = p.F, A = p.A;
  const size_t row = (size_t)b * F;
  float praw[PART ? NQ : 1][kMaxHeadSplit];
  float xin[NQ], tb[NQ], wvv[NQ], wp[NQ][kHeadMaxA];
  int fcl[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int f = lane + 64 * q;
    fcl[q] = f < F ? f : 0;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (PART) {
#pragma unroll
      for (int z = 0; z < kMaxHeadSplit; ++z)
        praw[q][z] = p.part[(size_t)(z < p.ksplit ? z : p.ksplit - 1) * p.part_stride + row + fcl[q]];
      tb[q] = p.tbias[fcl[q]];
    } else {
      xin[q] = p.feat[row + fcl[q]];
    }
    wvv[q] = p.wv[fcl[q]];
#pragma unroll
    for (int a = 0; a < kHeadMaxA; ++a) wp[q][a] = p.wpi[(size_t)fcl[q] * A + (a < A ? a : 0)];
  }
  const float mybias = p.bpi[lane < A ? lane : 0];
  const float bvv = p.bv[0];
  float fx[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const bool ok = lane + 64 * q < F;
    float x;
    if (PART) {
      float sacc = 0.f;
#pragma unroll
      for (int z = 0; z < kMaxHeadSplit; z += 4)
        sacc += ((z < p.ksplit ? praw[q][z] : 0.f) + (z + 1 < p.ksplit ? praw[q][z + 1] : 0.f)) +
                ((z + 2 < p.ksplit ? praw[q][z + 2] : 0.f) + (z + 3 < p.ksplit ? praw[q][z + 3] : 0.f));
      x = act_apply(sacc + tb[q], p.act_feat);
    } else {
      x = xin[q];
    }
    fx[q] = ok ? x : 0.f;
  }
  float acc[kHeadMaxA];
#pragma unroll
  for (int a = 0; a < kHeadMaxA; ++a) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) t = fmaf(fx[q], wp[q][a], t);
    acc[a] = t;
  }
  float sv = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) sv = fmaf(fx[q], wvv[q], sv);
  float mylogit = 0.f;
#pragma unroll
  for (int a = 0; a < kHeadMaxA; ++a) {
    const float t = wave_sum(acc[a]);
    if (a < A && lane == a) mylogit = t + mybias;
  }
  const float v = wave_sum(sv) + bvv;
  if (lane < A) p.logits[(size_t)b * A + lane] = mylogit;
  if (lane == 0) p.value[b] = v;
  if (PART) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int f = lane + 64 * q;
      if (f < F) p.feat_w[row + f] = fx[q];
    }
  }
}

int launch_impala_heads_fwd(const ImpalaHeadArgs& a, hipStream_t st) {
  if (a.A > kHeadMaxA || a.F > 512) return -1;
  const bool part = a.part != nullptr;
  if (part && a.ksplit > kMaxHeadSplit) return -1;
  const int nq = (a.F + 63) / 64;
  const dim3 grid(a.B), blk(64);
#define XT_IH(NQV)                                                                                   \
  do {                                                                                               \
    if (part) hipLaunchKernelGGL((impala_heads_fwd_kernel<NQV, true>), grid, blk, 0, st, a);         \
    else hipLaunchKernelGGL((impala_heads_fwd_kernel<NQV, false>), grid, blk, 0, st, a);             \
  } while (0)
  if (nq <= 1) XT_IH(1); else if (nq <= 2) XT_IH(2); else if (nq <= 4) XT_IH(4); else XT_IH(8);
#undef XT_IH
  XT_LAUNCH_CHECK();
  return 0;
}

// K2: v-trace + loss + d(logits, baseline) (the arithmetic of impala_loss_kernel, statement for statement) followed
// by the gradient w.r.t. the trunk features (the arithmetic of heads_dfeat_kernel).  Grid (trajectory, row block):
// every workgroup of a trajectory repeats the (cheap, LDS-resident) v-trace of the whole trajectory and then
// produces d(features) of its own kVtRows rows -- one workgroup per trajectory walking all T rows was a chain of
// T/4 dependent global-load round trips (47 us at T = 128 for breakout_impala's single trajectory).  The reverse scan (tf.scan,
// vtrace.py:94-106) is a wave-parallel suffix scan over affine maps (round 3, see below).
constexpr int kVtRows = 8;
template <int AM>
__global__ __launch_bounds__(256) void impala_vtrace_bwd_kernel(const ImpalaLossArgs p) {
  constexpr int MAXT = 256;
  __shared__ float s_delta[MAXT], s_dc[MAXT], s_val[MAXT + 1], s_vs[MAXT + 1], s_red[MAXT], s_dv[MAXT];
  __shared__ float s_dl[MAXT * AM];
  const int t = threadIdx.x;
  const int traj = blockIdx.x;
  const int T = p.T, A = p.A, F = p.F;
  const int Tm = T - 1;
  const size_t base = (size_t)traj * T;
  const bool lead = blockIdx.y == 0;             // the trajectory's first row block publishes d(heads) and the loss
  // the d(features) operands of this thread's first feature column do not depend on the v-trace: request them now,
  // their latency hides behind it
  const int r0 = blockIdx.y * kVtRows;
  float wpre[AM], xpre[kVtRows], wvpre;
  {
    const int f = t < F ? t : 0;
#pragma unroll
    for (int a = 0; a < AM; ++a) wpre[a] = p.wpi[(size_t)f * A + (a < A ? a : 0)];
    wvpre = p.wv[f];
#pragma unroll
    for (int u = 0; u < kVtRows; ++u) {
      const int r = r0 + u < T ? r0 + u : T - 1;
      xpre[u] = p.feat[(base + r) * F + f];
    }
  }
  float rho = 0.f, disc = 0.f, rew = 0.f, val = 0.f, ce = 0.f, ent = 0.f, logz = 0.f, mx = 0.f, z = 1.f, p_nval = 0.f;
  int act = 0;
  // every global operand of this thread's time step is requested before the first use: as rolled loops over the A
  // actions (runtime trip count) the logits were fetched one per iteration, each followed by s_waitcnt vmcnt(0) -- three
  // such loops = ~3 A dependent round trips at the head of a latency-bound kernel (ISA)
  float lgv[AM], blv[AM];
#pragma unroll
  for (int a = 0; a < AM; ++a) { lgv[a] = 0.f; blv[a] = 0.f; }
  if (t < T) s_val[t] = p.baseline[base + t];
  if (t < Tm) {
    const float* lg = p.logits + (base + t) * A;
    const float* bl = p.bp_logits + (base + t) * A;
#pragma unroll
    for (int a = 0; a < AM; ++a) {
      const int ac = a < A ? a : 0;
      lgv[a] = lg[ac];
      blv[a] = bl[ac];
    }
    act = p.action[base + t];
    const bool dn = p.done[base + t] != 0;
    const float rraw = p.reward[base + t];
    val = p.baseline[base + t];
    const float nval = p.baseline[base + t + 1];
    mx = lgv[0];
    float bmx = blv[0];
#pragma unroll
    for (int a = 1; a < AM; ++a)
      if (a < A) { mx = fmaxf(mx, lgv[a]); bmx = fmaxf(bmx, blv[a]); }
    z = 0.f;
    float bz = 0.f;
#pragma unroll
    for (int a = 0; a < AM; ++a)
      if (a < A) { z += expf(lgv[a] - mx); bz += expf(blv[a] - bmx); }
    logz = logf(z);
    float lga = lgv[0], bla = blv[0];
#pragma unroll
    for (int a = 1; a < AM; ++a)
      if (a == act) { lga = lgv[a]; bla = blv[a]; }
    const float tlp = (lga - mx) - logz;
    const float blp = (bla - bmx) - logf(bz);
    ce = -tlp;
    rho = expf(tlp - blp);
    disc = dn ? 0.f : p.gamma;
    rew = fminf(fmaxf(rraw, -1.f), 1.f);
    p_nval = nval;
#pragma unroll
    for (int a = 0; a < AM; ++a)
      if (a < A) {
        const float rl = lgv[a] - mx;
        ent += (expf(rl) / z) * (logz - rl);
      }
  }
  // The v-trace recurrence acc_t = delta_t + (discount_t c_t) acc_{t+1}, acc_Tm = 0 (tf.scan in reverse, vtrace.py:94-106)
  // is a composition of affine maps x -> B + A x, which is associative: a suffix scan over the pairs (A, B) with
  // (A1,B1) o (A2,B2) = (A1 A2, B1 + A1 B2) gives every acc_t in log2(64) shuffle steps inside a wave plus one
  // combination of the (at most four) wave totals, instead of a one-lane chain of T dependent LDS reads and FMAs
  // (round 2: 13.1 us for this kernel at T = 128).  Same real arithmetic, different rounding order than the serial
  // loop: values agree to ~1e-7 relative (the parity bar for vs / pg_adv is 1e-5); masks and indices are untouched.
  {
    float sa = (t < Tm) ? disc * fminf(1.f, rho) : 1.f;     // identity (1, 0) beyond the last transition
    float sb = (t < Tm) ? fminf(1.f, rho) * (rew + disc * p_nval - val) : 0.f;
    const int lane = t & 63, wv = t >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float oa = __shfl_down(sa, d, 64), ob = __shfl_down(sb, d, 64);
      if (lane + d < 64) { sb = fmaf(sa, ob, sb); sa *= oa; }
    }
    if (lane == 0) { s_delta[wv] = sa; s_dc[wv] = sb; }     // the wave's total map (s_delta / s_dc are scratch here)
    __syncthreads();
    float ta = sa, tb = sb;                                  // maps of the following waves, nearest first, applied to 0
    for (int w2 = wv + 1; w2 < 4; ++w2) { tb = fmaf(ta, s_dc[w2], tb); ta *= s_delta[w2]; }
    if (t < Tm) s_vs[t] = tb + val;
    if (t == Tm) s_vs[Tm] = s_val[Tm];
  }
  __syncthreads();
  float lterm = 0.f;
  if (t < Tm) {
    const float vs = s_vs[t], vsn = s_vs[t + 1];
    const float pg = fminf(1.f, rho) * (rew + disc * vsn - val);
#pragma unroll
    for (int a = 0; a < AM; ++a)
      if (a < A) {
        const float rl = lgv[a] - mx;
        const float pa = expf(rl) / z;
        const float lpa = rl - logz;
        const float onehot = (a == act) ? 1.f : 0.f;
        const float dl = pg * (pa - onehot) + 0.01f * (pa * (lpa + ent));
        if (lead) p.dlogits[(base + t) * A + a] = dl;
        s_dl[t * AM + a] = dl;
      }
    const float dvl = 0.5f * (val - vs);
    if (lead) p.dbaseline[base + t] = dvl;
    s_dv[t] = dvl;
    const float dvv = vs - val;
    lterm = ce * pg + 0.5f * (0.5f * dvv * dvv) + 0.01f * (-ent);
    if (lead && p.vs_out) p.vs_out[(size_t)traj * Tm + t] = vs;
    if (lead && p.pg_out) p.pg_out[(size_t)traj * Tm + t] = pg;
  } else if (t == Tm) {
    for (int a = 0; a < A; ++a) { if (lead) p.dlogits[(base + t) * A + a] = 0.f; s_dl[t * AM + a] = 0.f; }
    if (lead) p.dbaseline[base + t] = 0.f;
    s_dv[t] = 0.f;
  }
  {                                               // trajectory loss: wave sums, then the four of them in wave order
    const float ws = wave_sum(lterm);
    if ((t & 63) == 0) s_red[t >> 6] = ws;
  }
  __syncthreads();
  if (t == 0 && blockIdx.y == 0) p.traj_loss[traj] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  // d(features)[r, f] = (sum_a dlogits[r,a] Wpi[f,a] + dbaseline[r] Wv[f]) * act'(feature)   (heads_dfeat_kernel)
  // for this block's kVtRows rows: every row's feature load is issued before the first use
  for (int f = t; f < F; f += 256) {
    float w[AM], x[kVtRows];
    float wvf;
    if (f == t) {                                 // first feature column: loads issued at kernel entry
#pragma unroll
      for (int a = 0; a < AM; ++a) w[a] = wpre[a];
#pragma unroll
      for (int u = 0; u < kVtRows; ++u) x[u] = xpre[u];
      wvf = wvpre;
    } else {
#pragma unroll
      for (int a = 0; a < AM; ++a) w[a] = p.wpi[(size_t)f * A + (a < A ? a : 0)];
      wvf = p.wv[f];
#pragma unroll
      for (int u = 0; u < kVtRows; ++u) {
        const int r = r0 + u < T ? r0 + u : T - 1;
        x[u] = p.feat[(base + r) * F + f];
      }
    }
    float* dr = p.dfeat + base * F + f;
#pragma unroll
    for (int u = 0; u < kVtRows; ++u) {
      const int r = r0 + u;
      if (r < T) {
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < AM; ++a)
          if (a < A) s = fmaf(s_dl[r * AM + a], w[a], s);
        const float sv = s_dv[r] * wvf;
        dr[(size_t)r * F] = (s + sv) * act_grad(x[u], p.act_prev);
      }
    }
  }
}

int launch_impala_vtrace_bwd(const ImpalaLossArgs& a, int n_traj, hipStream_t st) {
  if (a.T > 256 || a.A > 32 || n_traj < 1) return -1;
  const dim3 grid(n_traj, (a.T + kVtRows - 1) / kVtRows);
  if (a.A <= 8) hipLaunchKernelGGL((impala_vtrace_bwd_kernel<8>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((impala_vtrace_bwd_kernel<32>), grid, dim3(256), 0, st, a);
  XT_LAUNCH_CHECK();
  return 0;
}

int launch_impala_loss_reduce(const float* traj_loss, int n, float* out, float* acc, hipStream_t st) {
  hipLaunchKernelGGL(impala_loss_reduce_kernel, dim3(1), dim3(64), 0, st, traj_loss, n, out, acc);
  XT_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- GAE (float64, bit-exact with numpy)
// one thread per trajectory; same association as xt/agent/ppo/ppo.py:92-104:
//   discount = (~done)*gamma ; delta = (reward + discount*next_v) - v ;
//   adv[j] += (adv[j+1]*discount[j])*lam ; target = adv + v.   Compiled with
//   -ffp-contract=off semantics via explicit __dmul_rn/__dadd_rn.
__global__ __launch_bounds__(64) void gae_f64_kernel(const float* __restrict__ value, const double* __restrict__ reward,
                                                     const uint8_t* __restrict__ done, double* __restrict__ adv,
                                                     double* __restrict__ target, float* __restrict__ old_value,
                                                     int n_traj, int T, double gamma, double lam) {
  const int tr = blockIdx.x * 64 + threadIdx.x;
  if (tr >= n_traj) return;
  const float* v = value + (size_t)tr * (T + 1);
  const double* r = reward + (size_t)tr * T;
  const uint8_t* d = done + (size_t)tr * T;
  double* ad = adv + (size_t)tr * T;
  double* tg = target + (size_t)tr * T;
  float* ov = old_value + (size_t)tr * T;
  double carry = 0.0;
  for (int j = T - 1; j >= 0; --j) {
    const double disc = d[j] ? 0.0 : gamma;               // ~done * GAMMA (bool*float -> 0.0 or GAMMA)
    const double vj = (double)v[j], vn = (double)v[j + 1];
    const double delta = __dsub_rn(__dadd_rn(r[j], __dmul_rn(disc, vn)), vj);
    double a = delta;
    if (j < T - 1) a = __dadd_rn(delta, __dmul_rn(__dmul_rn(carry, disc), lam));
    ad[j] = a;
    tg[j] = __dadd_rn(a, vj);
    ov[j] = v[j];
    carry = a;
  }
}

// The same arithmetic, one WORKGROUP per trajectory (T <= kGaeMaxT): the inputs are loaded by all threads at once,
// delta_j (no carried dependence) is computed per thread with the same explicit roundings, and ONLY the recurrence
// adv_j = delta_j + (adv_{j+1} * discount_j) * lam runs serially (thread 0, from LDS, chunks of 16 in registers);
// target = adv + v and the stores are parallel again.  The one-thread-per-trajectory form above pays one global-load
// round trip per time step: 42.7 us per 32 x 128 update against 14.6 us here.  Bit-for-bit the same results.
constexpr int kGaeMaxT = 1024;
__global__ __launch_bounds__(256) void gae_f64_block_kernel(const float* __restrict__ value, const double* __restrict__ reward,
                                                            const uint8_t* __restrict__ done, double* __restrict__ adv,
                                                            double* __restrict__ target, float* __restrict__ old_value,
                                                            int T, double gamma, double lam) {
  __shared__ double s_a[kGaeMaxT], s_dl[kGaeMaxT];     // delta, then adv ; discount * 1 (kept separately: (carry*disc)*lam)
  const int tr = blockIdx.x, t = threadIdx.x;
  const float* v = value + (size_t)tr * (T + 1);
  const double* r = reward + (size_t)tr * T;
  const uint8_t* d = done + (size_t)tr * T;
  for (int j = t; j < T; j += 256) {
    const double disc = d[j] ? 0.0 : gamma;
    const double vj = (double)v[j], vn = (double)v[j + 1];
    s_a[j] = __dsub_rn(__dadd_rn(r[j], __dmul_rn(disc, vn)), vj);
    s_dl[j] = disc;
  }
  __syncthreads();
  if (t == 0) {
    double carry = 0.0;
    for (int j0 = T; j0 > 0; j0 -= 16) {               // steps j0-1 .. j0-16, newest first
      double dl[16], dc[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int j = j0 - 1 - u;
        const int jc = j >= 0 ? j : 0;
        dl[u] = s_a[jc]; dc[u] = s_dl[jc];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int j = j0 - 1 - u;
        if (j >= 0) {
          double a = dl[u];
          if (j < T - 1) a = __dadd_rn(dl[u], __dmul_rn(__dmul_rn(carry, dc[u]), lam));
          s_a[j] = a;
          carry = a;
        }
      }
    }
  }
  __syncthreads();
  double* ad = adv + (size_t)tr * T;
  double* tg = target + (size_t)tr * T;
  float* ov = old_value + (size_t)tr * T;
  for (int j = t; j < T; j += 256) {
    const double a = s_a[j];
    const float vf = v[j];
    ad[j] = a;
    tg[j] = __dadd_rn(a, (double)vf);
    ov[j] = vf;
  }
}

// Ragged form for the learner-side ingest (ABI >= 9): the rollout's trajectories lie back to back as ROWS (trajectory i
// = rows [offsets[i], offsets[i+1])), value_rows[r] = V(s_r) -- which is also the reference's old_value column -- and
// boot[i] = V of the state after trajectory i's last step (value[T] of xt/agent/ppo/ppo.py:90).  Same arithmetic, same
// roundings, same serial recurrence as gae_f64_block_kernel; lengths differ per trajectory (CartPole: <= 200).
__global__ __launch_bounds__(256) void gae_f64_ragged_kernel(const float* __restrict__ value_rows, const float* __restrict__ boot,
                                                             const double* __restrict__ reward, const uint8_t* __restrict__ done,
                                                             const int32_t* __restrict__ offsets, double* __restrict__ adv,
                                                             double* __restrict__ target, double gamma, double lam) {
  __shared__ double s_a[kGaeMaxT], s_dl[kGaeMaxT];
  const int tr = blockIdx.x, t = threadIdx.x;
  const int r0 = offsets[tr], T = offsets[tr + 1] - r0;
  if (T <= 0) return;
  const float* v = value_rows + r0;
  const double* r = reward + r0;
  const uint8_t* d = done + r0;
  const float vb = boot[tr];
  if (T <= kGaeMaxT) {
    for (int j = t; j < T; j += 256) {
      const double disc = d[j] ? 0.0 : gamma;
      const double vj = (double)v[j], vn = (double)(j + 1 < T ? v[j + 1] : vb);
      s_a[j] = __dsub_rn(__dadd_rn(r[j], __dmul_rn(disc, vn)), vj);
      s_dl[j] = disc;
    }
    __syncthreads();
    if (t == 0) {
      double carry = 0.0;
      for (int j0 = T; j0 > 0; j0 -= 16) {
        double dl[16], dc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int j = j0 - 1 - u;
          const int jc = j >= 0 ? j : 0;
          dl[u] = s_a[jc]; dc[u] = s_dl[jc];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int j = j0 - 1 - u;
          if (j >= 0) {
            double a = dl[u];
            if (j < T - 1) a = __dadd_rn(dl[u], __dmul_rn(__dmul_rn(carry, dc[u]), lam));
            s_a[j] = a;
            carry = a;
          }
        }
      }
    }
    __syncthreads();
    for (int j = t; j < T; j += 256) {
      const double a = s_a[j];
      adv[r0 + j] = a;
      target[r0 + j] = __dadd_rn(a, (double)v[j]);
    }
  } else if (t == 0) {        // longer than the LDS form holds: one lane walks the trajectory
    double carry = 0.0;
    for (int j = T - 1; j >= 0; --j) {
      const double disc = d[j] ? 0.0 : gamma;
      const double vj = (double)v[j], vn = (double)(j + 1 < T ? v[j + 1] : vb);
      const double delta = __dsub_rn(__dadd_rn(r[j], __dmul_rn(disc, vn)), vj);
      double a = delta;
      if (j < T - 1) a = __dadd_rn(delta, __dmul_rn(__dmul_rn(carry, disc), lam));
      adv[r0 + j] = a;
      target[r0 + j] = __dadd_rn(a, vj);
      carry = a;
    }
  }
}

// zero-pad the innermost axis of a [rows, c_src] array to [rows, c_dst] (elements of 1 or 4 bytes): image observations
// whose channel count is not a multiple of 4 (examples/ant_ppo.yaml: [84, 84, 3]) enter the first layer as C = 4 with a
// zero plane -- the weights' fourth input-channel rows see a zero operand, get a zero gradient and stay zero (exact).
template <typename E>
__global__ __launch_bounds__(256) void pad_channels_kernel(const E* __restrict__ src, E* __restrict__ dst, long long rows,
                                                           int c_src, int c_dst, E fill) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * c_dst) return;
  const long long r = e / c_dst;
  const int c = (int)(e - r * c_dst);
  dst[e] = c < c_src ? src[r * c_src + c] : fill;
}

int launch_ppo_loss_gauss(const float* mean, const float* log_std, const float* value, int B, int A, const int32_t* idx,
                          const float* action, const float* old_logp, const double* adv, const float* old_v,
                          const double* target_v, float clip_ratio, float ent_coef, float vf_clip, float critic_coef,
                          float inv_b, float* dmean, float* dvalue, float* dls_rows, int ldls, float* terms,
                          hipStream_t st) {
  XT_REQUIRE(B > 0 && A > 0 && ldls >= A && mean && log_std && action && dls_rows, "xt_ppo_loss_gauss: bad arguments");
  hipLaunchKernelGGL(ppo_loss_gauss_kernel, dim3((B + 255) / 256), dim3(256), 0, st, mean, log_std, value, B, A, idx,
                     action, old_logp, adv, old_v, target_v, clip_ratio, ent_coef, vf_clip, critic_coef, inv_b, dmean,
                     dvalue, dls_rows, ldls, terms);
  XT_LAUNCH_CHECK();
  return 0;
}

}  // namespace xt

namespace xt {
// y = act(z), elementwise: the output of a layer whose pre-activation is kept (swish / gelu, see Layer::z_off)
__global__ __launch_bounds__(256) void act_apply_kernel(const float* __restrict__ z, float* __restrict__ y, long long count, int act) {
  const long long e4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e4 + 3 < count) {
    float4 v = *reinterpret_cast<const float4*>(z + e4);
    v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
    *reinterpret_cast<float4*>(y + e4) = v;
  } else {
    for (long long e = e4; e < count; ++e) y[e] = act_apply(z[e], act);
  }
}
int launch_act_apply(const float* z, float* y, long long count, int act, hipStream_t st) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(act_apply_kernel, dim3((unsigned)((count + 1023) / 1024)), dim3(256), 0, st, z, y, count, act);
  XT_LAUNCH_CHECK();
  return 0;
}
}  // namespace xt

extern "C" {

int xt_heads_fwd(const float* f_pi, const float* f_v, int32_t B, int32_t F, int32_t A, const float* wpi,
                 const float* bpi, const float* wv, const float* bv, float* logits, float* value, void* stream) {
  XT_REQUIRE(B > 0 && F > 0 && A > 0, "xt_heads_fwd: bad sizes");
  hipLaunchKernelGGL(xt::heads_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, xt::as_stream(stream), f_pi, f_v, B, F, A,
                     wpi, bpi, wv, bv, logits, value);
  XT_LAUNCH_CHECK();
  return 0;
}

int xt_ppo_loss(const float* logits, const float* value, int32_t B, int32_t A, const int32_t* idx,
                const int32_t* action, const float* old_logp, const double* adv, const float* old_v,
                const double* target_v, float clip_ratio, float ent_coef, float vf_clip, float critic_coef,
                float inv_b, float* dlogits, float* dvalue, float* loss_terms, void* stream) {
  XT_REQUIRE(B > 0 && A > 0, "xt_ppo_loss: bad sizes");
  hipLaunchKernelGGL(xt::ppo_loss_kernel, dim3((B + 255) / 256), dim3(256), 0, xt::as_stream(stream), logits, value, B, A,
                     idx, action, old_logp, adv, old_v, target_v, clip_ratio, ent_coef, vf_clip, critic_coef, inv_b,
                     dlogits, dvalue, loss_terms);
  XT_LAUNCH_CHECK();
  return 0;
}

int xt_ppo_loss_gauss(const float* mean, const float* log_std, const float* value, int32_t B, int32_t A,
                      const int32_t* idx, const float* action, const float* old_logp, const double* adv,
                      const float* old_v, const double* target_v, float clip_ratio, float ent_coef, float vf_clip,
                      float critic_coef, float inv_b, float* dmean, float* dvalue, float* dlogstd_rows,
                      float* loss_terms, void* stream) {
  return xt::launch_ppo_loss_gauss(mean, log_std, value, B, A, idx, action, old_logp, adv, old_v, target_v, clip_ratio,
                                   ent_coef, vf_clip, critic_coef, inv_b, dmean, dvalue, dlogstd_rows, A, loss_terms,
                                   xt::as_stream(stream));
}

namespace xt {
// One thread per sample (the minibatch of model.fit is 128 rows): softmax, the two loss terms of the sample and
// d loss / d logits through the softmax, d loss / d value.  terms[b] = (sum_a per-element policy loss, squared error).
__global__ __launch_bounds__(256) void keras_impala_loss_kernel(const float* __restrict__ logits, const float* __restrict__ value,
                                                                int B, int A, const int32_t* __restrict__ idx,
                                                                const float* __restrict__ adv, const float* __restrict__ onehot,
                                                                const float* __restrict__ target_v, float ent,
                                                                float* __restrict__ dlogits, float* __restrict__ dvalue,
                                                                float2* __restrict__ terms) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const long long row = idx ? (long long)idx[b] : (long long)b;
  const float* z = logits + (size_t)b * A;
  const float* y = onehot + (size_t)row * A;
  const float ad = adv[row];
  float mx = z[0];
  for (int a = 1; a < A; ++a) mx = fmaxf(mx, z[a]);
  float den = 0.f;
  for (int a = 0; a < A; ++a) den += expf(z[a] - mx);
  const float inv_den = 1.f / den, eps = 1e-10f, inv_ba = 1.f / ((float)B * (float)A);
  float lsum = 0.f, pg = 0.f;
  for (int a = 0; a < A; ++a) {
    const float p = expf(z[a] - mx) * inv_den;
    const float lp = logf(p + eps);
    lsum += ad * (-y[a] * lp) + ent * (p * lp);
    const float g = (-ad * y[a] / (p + eps) + ent * (lp + p / (p + eps))) * inv_ba;
    pg += p * g;
  }
  for (int a = 0; a < A; ++a) {
    const float p = expf(z[a] - mx) * inv_den;
    const float lp = logf(p + eps);
    const float g = (-ad * y[a] / (p + eps) + ent * (lp + p / (p + eps))) * inv_ba;
    dlogits[(size_t)b * A + a] = p * (g - pg);
  }
  const float diff = value[b] - target_v[row];
  dvalue[b] = diff / (float)B;
  terms[b] = make_float2(lsum, diff * diff);
}

// fixed-order reduction of the per-sample terms by one block
__global__ __launch_bounds__(256) void keras_impala_loss_reduce_kernel(const float2* __restrict__ terms, int B, int A,
                                                                       float* __restrict__ out, float* __restrict__ acc) {
  __shared__ double sp[256], sv[256];
  double p = 0.0, v = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) { p += (double)terms[b].x; v += (double)terms[b].y; }
  sp[threadIdx.x] = p; sv[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sp[threadIdx.x] += sp[threadIdx.x + o]; sv[threadIdx.x] += sv[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float lpi = (float)(sp[0] / ((double)B * (double)A)), mse = (float)(sv[0] / (double)B);
    const float l = lpi + 0.5f * mse;
    out[0] = l; out[1] = lpi; out[2] = mse;
    if (acc) { acc[0] += l * (float)B; acc[1] += (float)B; }
  }
}
// Advantage normalisation over a whole rollout, float64, in place:  adv <- (adv - mean(adv)) / (std(adv) + eps) with
// numpy's population std (two passes: mean, then mean((x - mean)^2)), the line the reference carries as a comment
// (xt/algorithm/ppo/ppo.py:73) -- an OPTION here (model_config ADV_NORM, default off).  One workgroup: every thread
// sums a strided slice, the 64 lanes of a wave combine with xor shuffles (DPP, no LDS), the 16 wave sums go through
// LDS in wave order: a fixed summation order, so the result is reproducible run to run (it differs from numpy's
// pairwise order in the last bits; the parity test states 1e-12).
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double block_sum_f64(double v, double* red) {
  v = wave_sum_f64(v);
  __syncthreads();                       // red may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}
__global__ __launch_bounds__(1024) void adv_normalize_f64_kernel(double* __restrict__ adv, long long n, double eps,
                                                                 double* __restrict__ stats) {
  __shared__ double red[16];
  double s = 0.0;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) s += adv[i];
  const double mean = block_sum_f64(s, red) / (double)n;
  double q = 0.0;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) { const double d = adv[i] - mean; q += d * d; }
  const double sd = sqrt(block_sum_f64(q, red) / (double)n);
  for (long long i = threadIdx.x; i < n; i += blockDim.x) adv[i] = (adv[i] - mean) / (sd + eps);
  if (stats && threadIdx.x == 0) { stats[0] = mean; stats[1] = sd; }
}

}  // namespace xt

int xt_keras_impala_loss(const float* logits, const float* value, int32_t B, int32_t A, const int32_t* idx,
                         const float* adv, const float* onehot, const float* target_v, float ent_coef,
                         float* dlogits, float* dvalue, float* out, float* acc, void* stream) {
  XT_REQUIRE(logits && value && adv && onehot && target_v && dlogits && dvalue && out, "xt_keras_impala_loss: null argument");
  XT_REQUIRE(B > 0 && A > 0 && A <= 64, "xt_keras_impala_loss: bad sizes (B=%d A=%d)", B, A);
  hipStream_t st = xt::as_stream(stream);
  float2* terms = reinterpret_cast<float2*>(out + 4);       // out: >= 4 + 2*B floats
  hipLaunchKernelGGL(xt::keras_impala_loss_kernel, dim3((B + 255) / 256), dim3(256), 0, st, logits, value, B, A, idx, adv,
                     onehot, target_v, ent_coef, dlogits, dvalue, terms);
  XT_LAUNCH_CHECK();
  hipLaunchKernelGGL(xt::keras_impala_loss_reduce_kernel, dim3(1), dim3(256), 0, st, terms, B, A, out, acc);
  XT_LAUNCH_CHECK();
  return 0;
}

int xt_ppo_loss_reduce(const float* loss_terms, int32_t B, float ent_coef, float critic_coef, float inv_b,
                       float* out, float* acc, void* stream) {
  hipLaunchKernelGGL(xt::ppo_loss_reduce_kernel, dim3(1), dim3(256), 0, xt::as_stream(stream), loss_terms, B, ent_coef,
                     critic_coef, inv_b, out, acc);
  XT_LAUNCH_CHECK();
  return 0;
}

int xt_impala_loss(const float* logits, const float* baseline, const float* bp_logits, const int32_t* action,
                   const uint8_t* done, const float* reward, int32_t n_traj, int32_t T, int32_t A, float gamma,
                   float* dlogits, float* dbaseline, float* out, float* acc, float* vs, float* pg_adv,
                   void* stream) {
  XT_REQUIRE(n_traj > 0 && T >= 2 && T <= 1024, "xt_impala_loss: need 2 <= T <= 1024 (got %d)", T);
  XT_REQUIRE(n_traj <= 4096, "xt_impala_loss: n_traj too large");
  // traj_loss scratch lives in out[4..4+n_traj)
  float* traj_loss = out + 4;
  hipStream_t st = xt::as_stream(stream);
  if (T <= 64)
    hipLaunchKernelGGL((xt::impala_loss_kernel<64>), dim3(n_traj), dim3(64), 0, st, logits, baseline, bp_logits, action,
                       done, reward, T, A, gamma, dlogits, dbaseline, traj_loss, vs, pg_adv);
  else if (T <= 128)
    hipLaunchKernelGGL((xt::impala_loss_kernel<128>), dim3(n_traj), dim3(128), 0, st, logits, baseline, bp_logits, action,
                       done, reward, T, A, gamma, dlogits, dbaseline, traj_loss, vs, pg_adv);
  else if (T <= 256)
    hipLaunchKernelGGL((xt::impala_loss_kernel<256>), dim3(n_traj), dim3(256), 0, st, logits, baseline, bp_logits, action,
                       done, reward, T, A, gamma, dlogits, dbaseline, traj_loss, vs, pg_adv);
  else
    hipLaunchKernelGGL((xt::impala_loss_kernel<1024>), dim3(n_traj), dim3(1024), 0, st, logits, baseline, bp_logits,
                       action, done, reward, T, A, gamma, dlogits, dbaseline, traj_loss, vs, pg_adv);
  XT_LAUNCH_CHECK();
  hipLaunchKernelGGL(xt::impala_loss_reduce_kernel, dim3(1), dim3(64), 0, st, traj_loss, n_traj, out, acc);
  XT_LAUNCH_CHECK();
  return 0;
}

int xt_heads_bwd(const float* f_pi, const float* f_v, int32_t B, int32_t F, int32_t A, const float* wpi,
                 const float* wv, const float* dlogits, const float* dvalue, int32_t act_prev, float* dwpi,
                 float* dbpi, float* dwv, float* dbv, float* df_pi, float* df_v, void* stream) {
  XT_REQUIRE(B > 0 && F > 0 && A > 0, "xt_heads_bwd: bad sizes");
  const int shared = (f_pi == f_v) ? 1 : 0;
  XT_REQUIRE(!shared || df_pi == df_v, "xt_heads_bwd: shared trunk needs df_pi == df_v");
  hipStream_t st = xt::as_stream(stream);
  hipLaunchKernelGGL(xt::heads_dfeat_kernel, dim3((B * F + 255) / 256), dim3(256), 0, st, f_pi, f_v, B, F, A, wpi, wv,
                     dlogits, dvalue, act_prev, shared, df_pi, df_v);
  XT_LAUNCH_CHECK();
  hipLaunchKernelGGL(xt::heads_wgrad_kernel, dim3((F + 63) / 64), dim3(256), 0, st, f_pi, f_v, B, F, A, dlogits, dvalue,
                     dwpi, dbpi, dwv, dbv);
  XT_LAUNCH_CHECK();
  return 0;
}

int xt_adv_normalize_f64(double* adv, int64_t n, double eps, double* stats, void* stream) {
  XT_REQUIRE(adv && n >= 0, "xt_adv_normalize_f64: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(xt::adv_normalize_f64_kernel, dim3(1), dim3(1024), 0, xt::as_stream(stream), adv, (long long)n, eps, stats);
  XT_LAUNCH_CHECK();
  return 0;
}

int xt_gae_f64(const float* value, const double* reward, const uint8_t* done, double* adv, double* target_value,
               float* old_value, int32_t n_traj, int32_t T, double gamma, double lam, void* stream) {
  XT_REQUIRE(n_traj >= 0 && T >= 0, "xt_gae_f64: bad sizes");
  if (n_traj == 0 || T == 0) return 0;
  if (T <= xt::kGaeMaxT)
    hipLaunchKernelGGL(xt::gae_f64_block_kernel, dim3(n_traj), dim3(256), 0, xt::as_stream(stream), value, reward, done,
                       adv, target_value, old_value, T, gamma, lam);
  else
    hipLaunchKernelGGL(xt::gae_f64_kernel, dim3((n_traj + 63) / 64), dim3(64), 0, xt::as_stream(stream), value, reward,
                       done, adv, target_value, old_value, n_traj, T, gamma, lam);
  XT_LAUNCH_CHECK();
  return 0;
}

int xt_gae_f64_ragged(const float* value_rows, const float* boot, const double* reward, const uint8_t* done,
                      const int32_t* offsets, double* adv, double* target_value, int32_t n_traj, double gamma, double lam,
                      void* stream) {
  XT_REQUIRE(n_traj >= 0, "xt_gae_f64_ragged: bad sizes");
  if (n_traj == 0) return 0;
  XT_REQUIRE(value_rows && boot && reward && done && offsets && adv && target_value, "xt_gae_f64_ragged: null argument");
  hipLaunchKernelGGL(xt::gae_f64_ragged_kernel, dim3(n_traj), dim3(256), 0, xt::as_stream(stream), value_rows, boot, reward,
                     done, offsets, adv, target_value, gamma, lam);
  XT_LAUNCH_CHECK();
  return 0;
}

int xt_pad_channels(const void* src, void* dst, int64_t rows, int32_t c_src, int32_t c_dst, int32_t elem_bytes,
                    int32_t fill_u8, void* stream) {
  XT_REQUIRE(src && dst && rows >= 0 && c_src > 0 && c_dst >= c_src && (elem_bytes == 1 || elem_bytes == 4),
             "xt_pad_channels: bad arguments (c_src=%d c_dst=%d elem_bytes=%d)", c_src, c_dst, elem_bytes);
  XT_REQUIRE(fill_u8 >= 0 && fill_u8 <= 255, "xt_pad_channels: fill byte %d outside [0,255]", fill_u8);
  if (rows == 0) return 0;
  const long long n = (long long)rows * c_dst;
  XT_REQUIRE((n + 255) / 256 < (1ll << 31), "xt_pad_channels: array too large");
  const dim3 grid((unsigned)((n + 255) / 256));
  if (elem_bytes == 1)
    hipLaunchKernelGGL((xt::pad_channels_kernel<uint8_t>), grid, dim3(256), 0, xt::as_stream(stream),
                       static_cast<const uint8_t*>(src), static_cast<uint8_t*>(dst), (long long)rows, c_src, c_dst,
                       (uint8_t)fill_u8);
  else
    hipLaunchKernelGGL((xt::pad_channels_kernel<float>), grid, dim3(256), 0, xt::as_stream(stream),
                       static_cast<const float*>(src), static_cast<float*>(dst), (long long)rows, c_src, c_dst, 0.f);
  XT_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
