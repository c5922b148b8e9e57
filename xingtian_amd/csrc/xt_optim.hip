// tf.clip_by_global_norm + TF1 AdamOptimizer over one flat fp32 buffer (HBM-bound).
// Replaces build_train_op, xt/model/ppo/ppo.py:97-102 and impala_cnn_opt.py:204-217.
// Three launches: per-block sum of squares (fixed order), single-block finalize
// (norm, clip scale, bias-corrected step size, beta powers), vectorised Adam update.
#include <stdlib.h>
#include <string.h>
#include "xt_common.h"
#include "xt_xgmi_dev.h"
#include <atomic>

namespace xt {

constexpr int kNormBlocks = 512;   // partial sums; scratch must hold >= kNormBlocks floats

__device__ void finalize_body(const float* partial, int nblocks, float clip_norm, float grad_scale, float lr,
                              float beta1, float beta2, int advance, float* state, const LossArgs& la, double* sh);

__device__ void loss_reduce_body(const LossArgs& la, double* sh);
__device__ double sqnorm_total_coherent(const float* partial, int nblocks, double* sh);
__device__ __forceinline__ void clip_scale(double sq, float clip_norm, float grad_scale, float* gnorm, float* scale);

// Grid barrier of the fused tail (all blocks of the launch are resident: the host checks).  Arrivals go through 64
// sub-counters (one 128-byte line each) and a top counter -- ~1500 same-address atomics would serialise at ~12 ns each --,
// the last arrival resets them and flips the sense word the others spin on (bounded: a launch that could not be
// co-resident would otherwise hang the GPU).
__device__ __forceinline__ void grid_arrive(unsigned int* counter, unsigned int* flag, unsigned int sense) {
  const unsigned nsub = gridDim.x < 64u ? gridDim.x : 64u;
  const unsigned sub = blockIdx.x % nsub;
  const unsigned cnt = (gridDim.x - sub + nsub - 1u) / nsub;
  if (__hip_atomic_fetch_add(counter + 32u * (1u + sub), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == cnt - 1u) {
    __hip_atomic_store(counter + 32u * (1u + sub), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsub - 1u) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(flag, sense ^ 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// -> false when the (bounded) wait ran out: the caller must NOT go on with partials / a step size it cannot trust
__device__ __forceinline__ bool grid_wait(unsigned int* flag, unsigned int sense) {
  for (unsigned it = 0; it < (1u << 24); ++it) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sense) return true;
    __builtin_amdgcn_s_sleep(2);
  }
  return false;
}
__device__ __forceinline__ void adam_advance(float* state, float lr, float beta1, float beta2);
__device__ void finalize_body(const float* partial, int nblocks, float clip_norm, float grad_scale, float lr,
                              float beta1, float beta2, int advance, float* state, const LossArgs& la, double* sh);

// data-parallel tail slots of THIS rank for one step: [rank] = rows, [16 + rank] = loss share, everything else zero
__device__ __forceinline__ float dp_tail_value(int i, int rank, float rows, float loss) {
  return i == rank ? rows : (i == kDpMaxWorld + rank ? loss : 0.f);
}

// the scatter ticket of the gradient-reduction launch (fused direct exchange): every block has pushed its float4s into the
// owners' inboxes and drained its stores; the LAST block of the launch raises this rank's ready flag at every peer.  Two-level
// block ticket (64 sub-counters + a top counter, one 128-byte line each, both re-armed by their last arriver): ~1700 blocks
// bumping one word would serialise at ~12 ns each -- measured round 6: 20 of the launch's 30 us with per-owner word tickets.
__device__ __forceinline__ void dp_scatter_ticket(const DpFinish& d, unsigned int* counter) {
  if (threadIdx.x != 0) return;
  const unsigned nsub = gridDim.x < 64u ? gridDim.x : 64u;
  const unsigned sub = blockIdx.x % nsub;
  const unsigned cnt = (gridDim.x - sub + nsub - 1u) / nsub;
  if (__hip_atomic_fetch_add(counter + 32u * (1u + sub), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != cnt - 1u) return;
  __hip_atomic_store(counter + 32u * (1u + sub), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != nsub - 1u) return;
  __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t seq = d.ctl[kCtlSeq] + 1;
  for (int q = 0; q < d.world; ++q)
    __hip_atomic_store(d.peers.flags[q] + kReadyOff + d.rank, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void grads_finish_kernel(const GradTable tab, float* __restrict__ partial,
                                                           const FinalizeArgs fin, const DpFinish dpf) {
  __shared__ float4 sh4[256];
  __shared__ float shs[256];
  __shared__ int s_last;
  __shared__ unsigned int s_sense;
  XT_TL(0);
  XT_TL_ROLE(60);
  // fused tail: the barrier flips a sense word; every block reads the old sense BEFORE it arrives (the flip needs all arrivals)
  unsigned int* const bar_flag = fin.counter ? fin.counter + 32u * 65u : nullptr;
  if (fin.enable == 3 && threadIdx.x == 0)
    s_sense = __hip_atomic_load(bar_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (fin.enable >= 2 && blockIdx.x == gridDim.x - 1) {
    // extra block of the "Adam computes the clip scale itself" form: everything of the old last-block finalize
    // that does not depend on the gradient norm (loss scalars, beta powers, step size) -- off the critical path
    loss_reduce_body(fin.loss, reinterpret_cast<double*>(sh4));
    if (threadIdx.x == 0) {
      adam_advance(fin.state, fin.lr_dev ? fin.lr_dev[0] : fin.lr, fin.beta1, fin.beta2);
      if (fin.enable == 3) {      // the step size crosses the barrier: write-through, drained, then arrive (and leave)
        __hip_atomic_store(fin.state + 3, fin.state[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        grid_arrive(fin.counter, bar_flag, s_sense);
      }
      shs[0] = fin.loss.out ? fin.loss.out[0] : 0.f;        // (thread 0 wrote it itself a few lines up)
    }
    if (dpf.tail || dpf.scatter) {
      // data-parallel tail: this rank's rows and loss share in its own two slots, zeros elsewhere (SUM = every rank's values)
      __syncthreads();
      const float loss = shs[0];
      if (!dpf.scatter) {
        if (threadIdx.x < kDpTailFloats) dpf.tail[threadIdx.x] = dp_tail_value(threadIdx.x, dpf.rank, dpf.rows, loss);
      } else {
        if (threadIdx.x < kDpTailFloats / 4) {
          const int i = 4 * (int)threadIdx.x;
          const long long v = dpf.nvec - kDpTailFloats / 4 + threadIdx.x;
          int64_t b;
          const int q = owner_of(v, dpf.nvec, dpf.world, b);
          reinterpret_cast<float4*>(dpf.peers.inbox_me[q])[v - b] =
              make_float4(dp_tail_value(i, dpf.rank, dpf.rows, loss), dp_tail_value(i + 1, dpf.rank, dpf.rows, loss),
                          dp_tail_value(i + 2, dpf.rank, dpf.rows, loss), dp_tail_value(i + 3, dpf.rank, dpf.rows, loss));
        }
        publish_fence();
        __syncthreads();
        dp_scatter_ticket(dpf, fin.counter);
      }
    }
    return;
  }
  int ei = 0;
  for (int q = 1; q < tab.n; ++q)
    if ((int)blockIdx.x >= tab.e[q].blk0) ei = q;
  const GradEntry& E = tab.e[ei];
  const int zl = E.zl, cols = 256 / zl;
  const int t = threadIdx.x, col = t % cols, z0 = t / cols;
  const int e0 = ((blockIdx.x - E.blk0) * cols + col) * 4;
  const bool active = e0 < E.count;
  const bool full = e0 + 3 < E.count;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    if (full) {
      // 8 slabs in flight per thread (clamped unconditional loads): the plain one-load-per-iteration loop spent
      // 74 % of its wave cycles in s_waitcnt
      for (int zb = z0; zb < E.nslab; zb += 8 * zl) {
        float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int z = zb + u * zl;
          q[u] = *reinterpret_cast<const float4*>(E.src + (size_t)(z < E.nslab ? z : z0) * E.stride + e0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (zb + u * zl < E.nslab) { a.x += q[u].x; a.y += q[u].y; a.z += q[u].z; a.w += q[u].w; }
        }
      }
    } else {
      for (int z = z0; z < E.nslab; z += zl) {
        const float* sp = E.src + (size_t)z * E.stride + e0;
        a.x += sp[0];
        if (e0 + 1 < E.count) a.y += sp[1];
        if (e0 + 2 < E.count) a.z += sp[2];
      }
    }
  }
  sh4[t] = a;
  __syncthreads();
  float sq = 0.f;
  if (z0 == 0 && active) {
    float4 r = sh4[col];
    for (int z = 1; z < zl; ++z) {
      const float4 q = sh4[z * cols + col];
      r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
    }
    float* dp = E.dst + e0;
    if (dpf.scatter) {
      // direct exchange fused into the step: the reduced float4 goes straight into its OWNER's inbox (posted remote write);
      // the lanes past the entry's end are alignment padding of the flat buffer: zeros
      if (!full) {
        if (e0 + 1 >= E.count) r.y = 0.f;
        if (e0 + 2 >= E.count) r.z = 0.f;
        r.w = 0.f;
      }
      const long long v = ((long long)(dp - dpf.grads_base)) >> 2;
      int64_t b;
      const int q = owner_of(v, dpf.nvec, dpf.world, b);
      reinterpret_cast<float4*>(dpf.peers.inbox_me[q])[v - b] = r;
      sq = r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
    } else if (full) {
      if (E.nslab > 1 || E.src != E.dst) *reinterpret_cast<float4*>(dp) = r;
      sq = r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
    } else {
      const bool wr = (E.nslab > 1 || E.src != E.dst);
      if (wr) dp[0] = r.x;
      sq = r.x * r.x;
      if (e0 + 1 < E.count) { if (wr) dp[1] = r.y; sq += r.y * r.y; }
      if (e0 + 2 < E.count) { if (wr) dp[2] = r.z; sq += r.z * r.z; }
    }
  }
  shs[t] = sq;
  XT_TL(1);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) shs[t] += shs[t + o];
    __syncthreads();
  }
  XT_TL(2);
  if (fin.enable == 3) {
    // ---- fused tail.  The thread that holds a reduced group of four elements also applies Adam to it, so only the
    // squared-norm partials (4 bytes per block, write-through) and the step size cross the grid barrier -- no gradient
    // is read from another block.  The parameter / moment loads are issued BEFORE the wait.
    const bool holder = (z0 == 0 && active);
    const long long off = (long long)(E.dst - fin.ap.grads) + e0;
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), mv = pv, vv = pv;
    if (holder) {
      if (full) {
        pv = *reinterpret_cast<const float4*>(fin.ap.params + off);
        mv = *reinterpret_cast<const float4*>(fin.ap.m + off);
        vv = *reinterpret_cast<const float4*>(fin.ap.v + off);
      } else {
        pv.x = fin.ap.params[off]; mv.x = fin.ap.m[off]; vv.x = fin.ap.v[off];
        if (e0 + 1 < E.count) { pv.y = fin.ap.params[off + 1]; mv.y = fin.ap.m[off + 1]; vv.y = fin.ap.v[off + 1]; }
        if (e0 + 2 < E.count) { pv.z = fin.ap.params[off + 2]; mv.z = fin.ap.m[off + 2]; vv.z = fin.ap.v[off + 2]; }
      }
    }
    float4 gr = sh4[col];      // (holders: the reduced gradient, recomputed from the LDS copies in the same order)
    if (holder) {
      for (int z = 1; z < zl; ++z) {
        const float4 q = sh4[z * cols + col];
        gr.x += q.x; gr.y += q.y; gr.z += q.z; gr.w += q.w;
      }
    }
    if (t == 0) {
      if (E.pre == 0) {
        __hip_atomic_store(partial + E.pblk0 + ((int)blockIdx.x - E.blk0), shs[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      grid_arrive(fin.counter, bar_flag, s_sense);
      s_last = grid_wait(bar_flag, s_sense) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) {
      // the barrier never completed (a grid that is not co-resident: the host-side check was wrong for this device):
      // leave parameters and moments untouched and raise the device-side error word the host reads with the loss
      if (t == 0) fin.state[6] = 1.f;
      return;
    }
    double* shd = reinterpret_cast<double*>(sh4);
    const double sqt = sqnorm_total_coherent(partial, fin.ap.npartials, shd);
    if (t == 0) {
      float gnorm, sc;
      clip_scale(sqt, fin.clip_norm, fin.grad_scale, &gnorm, &sc);
      shs[0] = sc;
      shs[1] = __hip_atomic_load(fin.state + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (blockIdx.x == 0) { fin.state[2] = sc; fin.state[4] = gnorm; }
    }
    __syncthreads();
    const float scale = shs[0], alpha = shs[1];
    const float omb1 = 1.f - fin.beta1, omb2 = 1.f - fin.beta2, eps = fin.ap.eps;
    if (holder) {
#define XT_ADAMF(c)                                   \
      {                                               \
        const float gg = gr.c * scale;                \
        mv.c += (gg - mv.c) * omb1;                   \
        vv.c += (gg * gg - vv.c) * omb2;              \
        pv.c -= (mv.c * alpha) / (sqrtf(vv.c) + eps); \
      }
      XT_ADAMF(x) XT_ADAMF(y) XT_ADAMF(z) XT_ADAMF(w)
#undef XT_ADAMF
      if (full) {
        *reinterpret_cast<float4*>(fin.ap.params + off) = pv;
        *reinterpret_cast<float4*>(fin.ap.m + off) = mv;
        *reinterpret_cast<float4*>(fin.ap.v + off) = vv;
      } else {
        fin.ap.params[off] = pv.x; fin.ap.m[off] = mv.x; fin.ap.v[off] = vv.x;
        if (e0 + 1 < E.count) { fin.ap.params[off + 1] = pv.y; fin.ap.m[off + 1] = mv.y; fin.ap.v[off + 1] = vv.y; }
        if (e0 + 2 < E.count) { fin.ap.params[off + 2] = pv.z; fin.ap.m[off + 2] = mv.z; fin.ap.v[off + 2] = vv.z; }
      }
    }
    XT_TL(4);
    return;
  }
  if (fin.enable != 1) {
    if (dpf.scatter) {            // (the local squared-norm partials are of no use: the norm is that of the EXCHANGED gradient)
      publish_fence();
      __syncthreads();
      dp_scatter_ticket(dpf, fin.counter);
      return;
    }
    if (t == 0) partial[E.pblk0 + ((int)blockIdx.x - E.blk0)] = shs[0];
    return;
  }
  // ---- last block to arrive finalises (norm, clip scale, Adam step size, loss scalars): saves a launch.
  // Publish/consume per the gfx950 recipe R1: the 4-byte partial is stored WRITE-THROUGH (relaxed agent-scope
  // atomic store = sc1, no per-block release fence / L2 write-back), drained with vmcnt(0), then a relaxed
  // agent ticket; only the last arriver does ONE agent acquire, then plain loads.
  if (t == 0) {
    __hip_atomic_store(partial + blockIdx.x, shs[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // two-level ticket: ~3000 same-address atomics would serialise (~12 ns each); 64 sub-counters (one 128-B line each) + 1 top
    const unsigned nsub = gridDim.x < 64u ? gridDim.x : 64u;
    const unsigned sub = blockIdx.x % nsub;
    const unsigned cnt = (gridDim.x - sub + nsub - 1u) / nsub;
    int last = 0;
    if (__hip_atomic_fetch_add(fin.counter + 32u * (1u + sub), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == cnt - 1u) {
      __hip_atomic_store(fin.counter + 32u * (1u + sub), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsub - 1u) {
        __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        last = 1;
      }
    }
    s_last = last;
  }
  __syncthreads();
  XT_TL(3);
  if (!s_last) return;
  finalize_body(partial, gridDim.x, fin.clip_norm, fin.grad_scale, fin.lr_dev ? fin.lr_dev[0] : fin.lr, fin.beta1, fin.beta2,
                1, fin.state, fin.loss, reinterpret_cast<double*>(sh4));
  XT_TL(4);
}

XT_TL_SETTER(optim)
#ifdef XT_TIMELINE
// launch-to-launch period of an (almost) empty kernel: the floor every dependent launch of the step pays
__global__ __launch_bounds__(256) void null_kernel(float* p, int wr) {
  if (wr && threadIdx.x == 0) p[blockIdx.x * 32] = 1.f;
}
// shader clock probe: s_memtime (shader cycles) against s_memrealtime (100 MHz) over a ~20 us spin on one wave
__global__ void clock_probe_kernel(unsigned long long* out) {
  const unsigned long long r0 = wall_clock64(), c0 = clock64();
  while (wall_clock64() - r0 < 2000ull) { }
  const unsigned long long r1 = wall_clock64(), c1 = clock64();
  if (threadIdx.x == 0) { out[0] = r1 - r0; out[1] = c1 - c0; }
}
extern "C" int xt_tl_clock_probe(unsigned long long* out, void* stream) {
  clock_probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out);
  return (int)hipGetLastError();
}
extern "C" int xt_tl_null_period(int reps, int nblocks, int wr, float* scratch, float* ms_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  null_kernel<<<nblocks, 256, 0, st>>>(scratch, wr);
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) null_kernel<<<nblocks, 256, 0, st>>>(scratch, wr);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / reps;
  hipEventDestroy(e0); hipEventDestroy(e1);
  return (int)hipGetLastError();
}
#endif

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long long count,
                                                             float* __restrict__ partial) {
  __shared__ float sh[256];
  const long long n4 = count >> 2;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    s += v * v;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// state: [0]=b1^t [1]=b2^t [2]=scale [3]=alpha [4]=gnorm [5]=step
// PPO loss scalars from the per-sample terms (fixed-order tree), xt/model/ppo/__init__.py
__device__ void loss_reduce_body(const LossArgs& la, double* sh) {
  if (!la.terms) {
    if (la.traj_loss && threadIdx.x == 0) {     // IMPALA: sum of the per-trajectory sums, trajectory order (float, as
      float s = 0.f;                            // impala_loss_reduce_kernel)
      for (int i = 0; i < la.n_traj; ++i) s += la.traj_loss[i];
      if (la.out) la.out[0] = s;
      if (la.acc) {
        if (la.acc_set) { la.acc[0] = s; la.acc[1] = 1.f; }
        else { la.acc[0] += s; la.acc[1] += 1.f; }
      }
    }
    return;
  }
  double t3[3] = {0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < la.B; b += 256) {
    t3[0] += (double)la.terms[(size_t)b * 4 + 0];
    t3[1] += (double)la.terms[(size_t)b * 4 + 1];
    t3[2] += (double)la.terms[(size_t)b * 4 + 2];
  }
  double tot[3];
  for (int q = 0; q < 3; ++q) {
    sh[threadIdx.x] = t3[q];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
      __syncthreads();
    }
    tot[q] = sh[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float surr = (float)tot[0] * la.inv_b, ent = (float)tot[1] * la.inv_b, vf = 0.5f * (float)tot[2] * la.inv_b;
    const float actor = -surr - la.ent_coef * ent;
    const float loss = actor + la.critic_coef * vf;
    if (la.out) { la.out[0] = loss; la.out[1] = actor; la.out[2] = vf; la.out[3] = ent; }
    if (la.acc) { la.acc[0] += loss; la.acc[1] += 1.f; }
  }
}

// sum of the per-block squared-norm partials (double, fixed order: identical in every block that calls it)
__device__ double sqnorm_total(const float* partial, int nblocks, double* sh) {
  double s = 0.0;
  // eight partials per thread in flight at once (clamped unconditional loads, same summation order as the plain loop:
  // as a rolled loop the ~8 dependent-latency round trips of a 2000-partial norm sat at the head of every Adam block)
  for (int base = threadIdx.x; base < nblocks; base += 256 * 8) {
    float q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + 256 * u;
      q[u] = partial[i < nblocks ? i : 0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (base + 256 * u < nblocks) s += (double)q[u];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  const double tot = sh[0];
  __syncthreads();
  return tot;
}
// the same sum with agent-scope loads: the partials were written (write-through) by other blocks of the SAME launch
__device__ double sqnorm_total_coherent(const float* partial, int nblocks, double* sh) {
  double s = 0.0;
  for (int base = threadIdx.x; base < nblocks; base += 256 * 8) {
    float q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + 256 * u;
      q[u] = __hip_atomic_load(partial + (i < nblocks ? i : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (base + 256 * u < nblocks) s += (double)q[u];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  const double tot = sh[0];
  __syncthreads();
  return tot;
}
// norm of (grad*grad_scale) and tf.clip_by_global_norm's factor: t * clip_norm * min(1/norm, 1/clip_norm)
__device__ __forceinline__ void clip_scale(double sq, float clip_norm, float grad_scale, float* gnorm, float* scale) {
  *gnorm = sqrtf((float)sq) * grad_scale;
  *scale = clip_norm * fminf(1.f / *gnorm, 1.f / clip_norm) * grad_scale;
}
// Adam step-size bookkeeping that does not depend on the gradient: beta powers, lr_t, step counter
__device__ __forceinline__ void adam_advance(float* state, float lr, float beta1, float beta2) {
  const float b1p = state[0] * beta1, b2p = state[1] * beta2;
  state[0] = b1p; state[1] = b2p;
  state[3] = lr * sqrtf(1.f - b2p) / (1.f - b1p);
  state[5] += 1.f;
}

__device__ void finalize_body(const float* partial, int nblocks, float clip_norm, float grad_scale, float lr,
                              float beta1, float beta2, int advance, float* state, const LossArgs& la, double* sh) {
  loss_reduce_body(la, sh);
  const double sq = sqnorm_total(partial, nblocks, sh);
  if (threadIdx.x == 0) {
    float gnorm, sc;
    clip_scale(sq, clip_norm, grad_scale, &gnorm, &sc);
    state[2] = sc;
    state[4] = gnorm;
    if (advance) adam_advance(state, lr, beta1, beta2);
  }
}

__global__ __launch_bounds__(256) void norm_finalize_kernel(const float* __restrict__ partial, int nblocks, float clip_norm,
                                                            float grad_scale, float lr, float beta1, float beta2,
                                                            int advance, float* __restrict__ state, LossArgs la,
                                                            const float* __restrict__ lr_dev) {
  __shared__ double sh[256];
  finalize_body(partial, nblocks, clip_norm, grad_scale, lr_dev ? lr_dev[0] : lr, beta1, beta2, advance, state, la, sh);
}

__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v, long long count,
                                                      float beta1, float beta2, float eps, const float* __restrict__ state) {
  const float scale = state[2], alpha = state[3];
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  const long long n4 = count >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float4 pv = reinterpret_cast<float4*>(p)[i];
#define XT_ADAM1(c)                                   \
    {                                                 \
      const float gg = gv.c * scale;                  \
      mv.c += (gg - mv.c) * omb1;                     \
      vv.c += (gg * gg - vv.c) * omb2;                \
      pv.c -= (mv.c * alpha) / (sqrtf(vv.c) + eps);   \
    }
    XT_ADAM1(x) XT_ADAM1(y) XT_ADAM1(z) XT_ADAM1(w)
#undef XT_ADAM1
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
    reinterpret_cast<float4*>(p)[i] = pv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float gg = g[i] * scale;
    float mm = m[i], vv = v[i];
    mm += (gg - mm) * omb1;
    vv += (gg * gg - vv) * omb2;
    m[i] = mm; v[i] = vv;
    p[i] -= (mm * alpha) / (sqrtf(vv) + eps);
  }
}

// Adam + global-norm clip where EVERY block derives the clip factor itself from the squared-norm partials of
// grads_finish_kernel (1920 floats from L2, fixed-order double sum -> bitwise the same factor in every block):
// removes the serial "last block finalises" tail (ticket + 6 us single-block reduction) from the step.  Block 0
// also publishes scale / gnorm to state[2] / state[4] (nobody reads them inside this launch).
// ---- the data-parallel part of an optimiser launch (DpStep)
// begin: returns false when the comm's sticky error word is set (a
// bounded wait ran out, now or earlier): the update is then SKIPPED -- parameters and slots stay those of the last good step
// instead of absorbing whatever arrived -- and the bits travel to the host in loss_acc[2]
__device__ bool dp_step_begin(const DpStep& dp) {
  __shared__ int s_ok;
  if (!dp.flags) return true;
  // (no wait here: dp_reduce_wait_kernel's last block has seen every rank's done flag before this launch started -- an
  // optimiser launch whose 512 workgroups all spin on another process's flags starved that process's kernels when the
  // ranks share one GPU: measured round 6, two ranks x B = 320: 3 s time-outs; the reduce launch's <= 128 small
  // workgroups are the footprint that has run since round 5)
  if (threadIdx.x == 0)
    s_ok = __hip_atomic_load(dp.ctl + kCtlErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u ? 1 : 0;
  __syncthreads();
  return s_ok != 0;
}

// The exchange launch of the step that has the direct exchange fused in (xt_net_set_direct).  The scatter was done by the
// gradient-reduction kernel (every reduced float4 written straight into its owner's inbox), the gather is done by the
// optimiser kernel (reads the result buffer).  <= 128 small workgroups: (1) wait for every rank's scatter, sum my slice
// over the inbox slots in FIXED rank order 0..N-1, push it to every peer's result buffer and leave the squared norm of
// each workgroup's share (gradient part only: the tail slots carry rows / losses) in every peer's norm_part[rank][block],
// raise my done flag everywhere; (2) wait until every rank's done flag has arrived -- so that the optimiser launch behind
// this one starts with the whole reduced gradient and all partials in place and never spins.
__global__ void __launch_bounds__(256) dp_reduce_wait_kernel(const DpStep dp) {
  __shared__ float s_sq[256];
  const uint32_t seq = dp.ctl[kCtlSeq] + 1;
  wait_all(dp.flags, kReadyOff, dp.world, seq, dp.ctl, dp.timeout_ticks, kErrScatterWait);
  int64_t b, e;
  slice_of(dp.nvec, dp.rank, dp.world, b, e);
  const int64_t per = ((e - b) + gridDim.x - 1) / gridDim.x;
  const int64_t lo = b + (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < e ? lo + per : e;
  float sq = 0.f;
  // four vectors per thread in flight (clamped unconditional loads): the inbox is UNCACHED memory, every load is a full
  // memory round trip
  for (int64_t v0 = lo + threadIdx.x; v0 < hi; v0 += 4 * 256) {
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t v = v0 + u * 256 < hi ? v0 + u * 256 : v0;
      acc[u] = *reinterpret_cast<const float4*>(dp.inbox + (v - b) * 4);
    }
    for (int p = 1; p < dp.world; ++p) {          // FIXED order 0, 1, ..., N-1: the sum is a function of the data only
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t v = v0 + u * 256 < hi ? v0 + u * 256 : v0;
        x[u] = *reinterpret_cast<const float4*>(dp.inbox + p * dp.slice_cap + (v - b) * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc[u].x += x[u].x; acc[u].y += x[u].y; acc[u].z += x[u].z; acc[u].w += x[u].w; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t v = v0 + u * 256;
      if (v < hi) {
        for (int p = 0; p < dp.world; ++p) reinterpret_cast<float4*>(dp.peers.result[p])[v] = acc[u];
        if (v < dp.nvec_grad)      // (not the tail slots)
          sq += acc[u].x * acc[u].x + acc[u].y * acc[u].y + acc[u].z * acc[u].z + acc[u].w * acc[u].w;
      }
    }
  }
  s_sq[threadIdx.x] = sq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_sq[threadIdx.x] += s_sq[threadIdx.x + o];
    __syncthreads();
  }
  if ((int)threadIdx.x < dp.world) dp.peers.norm_part[threadIdx.x][dp.rank * gridDim.x + blockIdx.x] = s_sq[0];
  publish_fence();
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(dp.ctl + kCtlRed, 1u) == gridDim.x - 1u) {
    for (int p = 0; p < dp.world; ++p)
      __hip_atomic_store(dp.peers.flags[p] + kDoneOff + dp.rank, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  wait_all(dp.flags, kDoneOff, dp.world, seq, dp.ctl, dp.timeout_ticks, kErrReduceWait);
}

int launch_dp_reduce_wait(const DpStep* dp, hipStream_t st) {
  XT_REQUIRE(dp && dp->flags && dp->red_blocks >= 1 && dp->red_blocks <= kDpRedBlocksMax, "dp_reduce_wait: bad arguments");
  hipLaunchKernelGGL(dp_reduce_wait_kernel, dim3(dp->red_blocks), dim3(256), 0, st, *dp);
  XT_LAUNCH_CHECK();
  return 0;
}

// block 0: the global loss (every rank's share, rank order -> the same bits everywhere) and the row check
__device__ void dp_tail_consume(const DpStep& dp, bool ok) {
  if (!dp.tail || blockIdx.x != 0 || threadIdx.x != 0) return;
  uint32_t err = 0u;
  if (ok) {
    float s = 0.f;
    const float rows0 = dp.tail[0];
    for (int r = 0; r < dp.world; ++r) {
      s += dp.tail[kDpMaxWorld + r];
      if (dp.tail[r] != rows0) err = kErrRowsMismatch;
    }
    if (dp.acc) { dp.acc[0] += dp.loss_scale * s; dp.acc[1] += 1.f; }
    if (err && dp.ctl) atomicOr(dp.ctl + kCtlErr, err);
  }
  if (dp.ctl) err |= __hip_atomic_load(dp.ctl + kCtlErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (err && dp.acc) dp.acc[2] = (float)err;
}
// end: direct exchange -> the last block re-arms the comm for the next step (tickets, sequence number)
__device__ void dp_step_end(const DpStep& dp) {
  if (!dp.flags) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(dp.ctl + kCtlGat, 1u) == gridDim.x - 1) {
      const uint32_t seq = dp.ctl[kCtlSeq] + 1;
      dp.ctl[kCtlRed] = 0;
      dp.ctl[kCtlGat] = 0;
      for (int q = 0; q < dp.world; ++q) dp.ctl[kCtlScat + q] = 0;
      __threadfence();
      dp.ctl[kCtlSeq] = seq;
    }
  }
}

// IO (xt_train_io.tail_in_graph = 2 with the tail folded in, IoFold in xt_common.h): block 0 first reports the train's loss to
// the host through the mailbox (the sums are final before the optimiser runs), every block also writes its updated parameters
// to the snapshot buffer with system-scope write-through stores (the SDMA engine that copies it out reads memory, not an XCD's
// L2), and the last block to finish reports the snapshot.  IO = false is the kernel of every other path, unchanged.
template <bool IO>
__global__ __launch_bounds__(256) void adam_tf_clip_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, long long count,
                                                           float beta1, float beta2, float eps, float* __restrict__ state,
                                                           const float* __restrict__ partial, int nblocks,
                                                           float clip_norm, float grad_scale, const DpStep dp, const IoFold io) {
  __shared__ double sh[256];
  __shared__ float s_scale;
  const bool ok = dp_step_begin(dp);
  dp_tail_consume(dp, ok);
  if (!ok) { dp_step_end(dp); return; }
  if (IO && blockIdx.x == 0) {
    const int t = threadIdx.x;
    uint32_t seq = 0;
    if (t == 0) {
      const unsigned long long dst = __hip_atomic_load(&io.mb->publish_dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      seq = __hip_atomic_load(&io.mb->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(io.fwd + 0, dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(io.fwd + 1, (unsigned long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t < 4) {
      const float val = io.acc[t];
      __hip_atomic_store(&io.mb->loss[t], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      io.loss_out[t] = val;
      io.acc[t] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t == 0) __hip_atomic_store(&io.mb->loss_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const double sq = sqnorm_total(partial, nblocks, sh);
  if (threadIdx.x == 0) {
    float gnorm, sc;
    clip_scale(sq, clip_norm, grad_scale, &gnorm, &sc);
    s_scale = sc;
    if (blockIdx.x == 0) { state[2] = sc; state[4] = gnorm; }
  }
  __syncthreads();
  const float scale = s_scale, alpha = state[3];
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  const long long n4 = count >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float4 pv = reinterpret_cast<float4*>(p)[i];
#define XT_ADAM1(c)                                   \
    {                                                 \
      const float gg = gv.c * scale;                  \
      mv.c += (gg - mv.c) * omb1;                     \
      vv.c += (gg * gg - vv.c) * omb2;                \
      pv.c -= (mv.c * alpha) / (sqrtf(vv.c) + eps);   \
    }
    XT_ADAM1(x) XT_ADAM1(y) XT_ADAM1(z) XT_ADAM1(w)
#undef XT_ADAM1
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
    reinterpret_cast<float4*>(p)[i] = pv;
    if (IO) {
#if defined(__HIP_DEVICE_COMPILE__)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(io.snap, 0, 0x7fffffff, 0x00020000);
      xt_u32x4 q;
      q.x = __float_as_uint(pv.x); q.y = __float_as_uint(pv.y); q.z = __float_as_uint(pv.z); q.w = __float_as_uint(pv.w);
      // sc1: written THROUGH the L2 to memory (what the SDMA engine reads) like every large output of this library; the
      // system-scope form (sc0 sc1) made this kernel 21.5 us instead of 7.4 + 8 for the separate snapshot kernel
      __builtin_amdgcn_raw_buffer_store_b128(q, rs, (int)(i * 16), 0, kAuxSc1);
#endif
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float gg = g[i] * scale;
    float mm = m[i], vv = v[i];
    mm += (gg - mm) * omb1;
    vv += (gg * gg - vv) * omb2;
    m[i] = mm; v[i] = vv;
    const float pn = p[i] - (mm * alpha) / (sqrtf(vv) + eps);
    p[i] = pn;
    if (IO) __hip_atomic_store(io.snap + i, pn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (IO) {
    // this block's snapshot stores are acknowledged; the last block to get here reports the snapshot (the sequence number
    // through device memory: the host may have rewritten the mailbox for the next train since block 0 reported the loss)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      // (no fence: block 0's agent-scope stores to fwd were acknowledged before ITS ticket, the load below is agent-scope too;
      // a __threadfence() per block is an L2 write-back per block -- it made this kernel 24 us)
      unsigned int* ticket = reinterpret_cast<unsigned int*>(io.fwd + 2);
      if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long seq = __hip_atomic_load(io.fwd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&io.mb->snap_seq, (uint32_t)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  dp_step_end(dp);
}

// tf.train.RMSPropOptimizer(centered=True, momentum=0) after tf.clip_by_global_norm, TF1 apply_centered_rms_prop:
//   ms = decay*ms + (1-decay)*g^2;  mg = decay*mg + (1-decay)*g;  var -= lr * g / sqrt(ms - mg^2 + epsilon)
// (impala_cnn_opt.py:205-215).  Same structure as adam_tf_clip_kernel: every block derives the clip factor itself.
__global__ __launch_bounds__(256) void rmsprop_tf_clip_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                              float* __restrict__ mg, float* __restrict__ ms,
                                                              long long count, float lr_arg, float decay, float eps,
                                                              float* __restrict__ state, const float* __restrict__ partial,
                                                              int nblocks, float clip_norm, float grad_scale,
                                                              const float* __restrict__ lr_dev, const DpStep dp) {
  __shared__ double sh[256];
  __shared__ float s_scale;
  const float lr = lr_dev ? lr_dev[0] : lr_arg;
  const bool ok = dp_step_begin(dp);
  dp_tail_consume(dp, ok);
  if (!ok) { dp_step_end(dp); return; }
  const double sq = sqnorm_total(partial, nblocks, sh);
  if (threadIdx.x == 0) {
    float gnorm, sc;
    clip_scale(sq, clip_norm, grad_scale, &gnorm, &sc);
    s_scale = sc;
    if (blockIdx.x == 0) { state[2] = sc; state[4] = gnorm; }
  }
  __syncthreads();
  const float scale = s_scale, omd = 1.f - decay;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    const float gg = g[i] * scale;
    float a = ms[i], b = mg[i];
    a += (gg * gg - a) * omd;
    b += (gg - b) * omd;
    ms[i] = a; mg[i] = b;
    p[i] -= lr * gg / sqrtf(a - b * b + eps);
  }
  dp_step_end(dp);
}

__global__ void adam_state_init_kernel(float* state) {
  if (threadIdx.x < 8) state[threadIdx.x] = (threadIdx.x < 2) ? 1.f : 0.f;
}

// entries: blk0/nblk/zl are filled here.  partial needs room for the returned block count.
// `select` (bit i = entry i of the table is reduced by THIS launch; 0 = all).  The slot of every squared-norm partial
// is that of the FULL table, so a table reduced by two partial launches leaves the same partial array (bitwise the
// same norm) as one launch.  `early` = the entries whose slots come first: a partial launch of exactly these entries
// may run before the slab counts of the others are known (their slots do not depend on them).
int launch_grads_finish(GradTable* tab, float* partial, int max_partials, int* nblocks_out, const FinalizeArgs* fin,
                        hipStream_t st, unsigned select, unsigned early, const DpFinish* dpf) {
  int blk = 0;
  for (int pass = 0; pass < 2; ++pass)
  for (int i = 0; i < tab->n; ++i) {
    if ((((early >> i) & 1u) != 0) != (pass == 0)) continue;
    GradEntry& E = tab->e[i];
    int zl = 1;
    // deep entries (the first layer's one-slab-per-workgroup gradients: 250 slabs) get up to 32 z lanes: their blocks
    // were the stragglers of the launch (31 dependent slab loads per lane, 64 blocks) while ~800 shallow blocks had finished
    const int zcap = (tuning().reduce_deep_lanes > 0 && E.nslab >= tuning().reduce_deep_lanes) ? 32 : tuning().reduce_z_lanes;
    while (zl < E.nslab && zl < zcap) zl <<= 1;   // fewer z lanes = longer contiguous runs per wave (512 B at 8)
    E.zl = zl;
    const int cols = 256 / zl;
    E.pblk0 = blk;
    E.nblk = ((E.count + 3) / 4 + cols - 1) / cols;
    blk += E.pre > 0 ? E.pre : E.nblk;        // (a pre-reduced entry owns as many slots as its producer had blocks)
    XT_REQUIRE((((uintptr_t)E.src | (uintptr_t)E.dst) & 15) == 0 && (E.stride % 4) == 0,
               "grads_finish: entry %d not 16-byte aligned", i);
  }
  XT_REQUIRE(blk > 0 && blk <= max_partials, "grads_finish: %d partial blocks > scratch %d", blk, max_partials);
  FinalizeArgs f;
  if (fin) f = *fin; else { memset(&f, 0, sizeof(f)); }
  const bool fused = (f.enable == 3);
  GradTable sub;
  sub.n = 0;
  int grid = 0;
  bool any_pre = false;
  for (int i = 0; i < tab->n; ++i) {
    if (select && !((select >> i) & 1u)) continue;
    if (tab->e[i].pre > 0) {
      any_pre = true;
      // reduced and squared by its producer; the fused tail still has to update it, the direct exchange still has to push it
      if (!fused && !(dpf && dpf->scatter)) continue;
    }
    GradEntry& E = sub.e[sub.n++];
    E = tab->e[i];
    E.blk0 = grid;
    grid += E.nblk;
  }
  XT_REQUIRE(grid > 0, "grads_finish: empty selection");
  XT_REQUIRE(f.enable != 1 || (!any_pre && !select),
             "grads_finish: the last-block finalize form needs the whole table in one launch (no partial launches, no "
             "pre-reduced entries)");
  XT_REQUIRE(!fused || (!select && f.counter), "grads_finish: the fused tail takes the whole table and a barrier scratch");
  if (fused) f.ap.npartials = blk;
  DpFinish d;
  if (dpf) d = *dpf; else memset(&d, 0, sizeof(d));
  if (d.tail || d.scatter)
    XT_REQUIRE(f.enable == 2 && !select, "grads_finish: the data-parallel tail rides in the extra block of the whole-table launch");
  if (d.scatter) {
    XT_REQUIRE(d.world >= 1 && d.world <= kDpMaxWorld && d.grads_base && d.ctl && f.counter &&
               d.nvec >= kDpTailFloats / 4 + d.world, "grads_finish: bad direct-exchange arguments");
    for (int i = 0; i < sub.n; ++i) {
      const long long off = (long long)(sub.e[i].dst - d.grads_base);
      XT_REQUIRE(off >= 0 && off % 4 == 0 && off + sub.e[i].count <= (d.nvec - kDpTailFloats / 4) * 4,
                 "grads_finish: entry %d lies outside the exchanged buffer", i);
    }
  }
  hipLaunchKernelGGL(grads_finish_kernel, dim3(grid + (f.enable >= 2 ? 1 : 0)), dim3(256), 0, st, sub, partial, f, d);
  XT_LAUNCH_CHECK();
  *nblocks_out = blk;
  return 0;
}

// how many workgroups of the fused tail can be resident at once on this device (its grid barrier needs all of them)
int grads_finish_resident_blocks() {
  static std::atomic<int> cached_dev[64];      // per device: a process may drive several (different) GPUs
  int per_cu = 0, cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::atomic<int>& cached = cached_dev[dev & 63];
  int v = cached.load();
  if (v > 0) return v;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, grads_finish_kernel, 256, 0) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  v = per_cu * cus;
  cached.store(v);
  return v;
}
// blocks the fused tail would launch for this table (pre-reduced entries included, + the scalar block)
int grads_finish_fused_grid(const GradTable* tab) {
  int grid = 1;
  for (int i = 0; i < tab->n; ++i) {
    const GradEntry& E = tab->e[i];
    int zl = 1;
    const int zcap = (tuning().reduce_deep_lanes > 0 && E.nslab >= tuning().reduce_deep_lanes) ? 32 : tuning().reduce_z_lanes;
    while (zl < E.nslab && zl < zcap) zl <<= 1;
    const int cols = 256 / zl;
    grid += ((E.count + 3) / 4 + cols - 1) / cols;
  }
  return grid;
}

int launch_rmsprop_clip(float* param, const float* grad, float* mg, float* ms, long long count, float lr, float decay,
                        float eps, float* state, const float* partial, int nblocks, float clip_norm, float grad_scale,
                        hipStream_t st, const float* lr_dev, const DpStep* dp, int block_cap) {
  int nb = (int)((count + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (block_cap > 0 && nb > block_cap) nb = block_cap;      // (blocks that wait for other ranks must all be resident)
  if (nb < 1) nb = 1;
  DpStep d;
  if (dp) d = *dp; else memset(&d, 0, sizeof(d));

  hipLaunchKernelGGL(rmsprop_tf_clip_kernel, dim3(nb), dim3(256), 0, st, param, grad, mg, ms, count, lr, decay, eps,
                     state, partial, nblocks, clip_norm, grad_scale, lr_dev, d);
  XT_LAUNCH_CHECK();
  return 0;
}

int launch_adam_clip(float* param, const float* grad, float* m, float* v, long long count, float beta1, float beta2,
                     float eps, float* state, const float* partial, int nblocks, float clip_norm, float grad_scale,
                     hipStream_t st, const DpStep* dp, int block_cap, const IoFold* io) {
  XT_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
             "adam: buffers must be 16-byte aligned");
  int nb = (int)((count / 4 + 255) / 256);
  // at most 512 blocks, two per CU in one dispatch round: every block re-derives the clip factor from the ~1 700
  // squared-norm partials before it touches an element, so fewer, fatter blocks repeat that chain less often and the grid
  // dispatches faster (round 4, same-box A/B per 52-step update: 828 blocks 6.810 / 6.808 ms, 512: 6.769 / 6.776, 256:
  // 6.789 / 6.797).  Element-wise arithmetic: bitwise the same parameters.
  if (nb > 512) nb = 512;
  if (block_cap > 0 && nb > block_cap) nb = block_cap;      // (blocks that wait for other ranks must all be resident)
  if (nb < 1) nb = 1;
  DpStep d;
  if (dp) d = *dp; else memset(&d, 0, sizeof(d));

  IoFold f;
  memset(&f, 0, sizeof(f));
  if (io) {
    XT_REQUIRE(count * 4 < 0x7fffffffLL, "adam: %lld parameters exceed the snapshot store's 2 GiB offset range", count);
    f = *io;
    hipLaunchKernelGGL(adam_tf_clip_kernel<true>, dim3(nb), dim3(256), 0, st, param, grad, m, v, count, beta1, beta2, eps,
                       state, partial, nblocks, clip_norm, grad_scale, d, f);
  } else {
    hipLaunchKernelGGL(adam_tf_clip_kernel<false>, dim3(nb), dim3(256), 0, st, param, grad, m, v, count, beta1, beta2, eps,
                       state, partial, nblocks, clip_norm, grad_scale, d, f);
  }
  XT_LAUNCH_CHECK();
  return 0;
}

}  // namespace xt

// ------------------------------------------------------------------ tf.keras Adam with per-tensor clipnorm
namespace xt {
constexpr int kKerasSlices = 16;      // squared-norm partials per tensor (fixed-order sum -> reproducible)
constexpr int kKerasMaxSeg = 32;
struct KerasSegs {
  int n;
  long long off[kKerasMaxSeg], size[kKerasMaxSeg];
};

// block (slice, tensor): sum of squares of its slice of the tensor's gradient
__global__ __launch_bounds__(256) void keras_seg_sqnorm_kernel(const float* __restrict__ g, const KerasSegs segs,
                                                               float* __restrict__ partial) {
  __shared__ double sh[256];
  const int sg = blockIdx.y, sl = blockIdx.x;
  const long long n = segs.size[sg], per = (n + kKerasSlices - 1) / kKerasSlices;
  const long long lo = sl * per, hi = lo + per < n ? lo + per : n;
  const float* p = g + segs.off[sg];
  double acc = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) { const double x = (double)p[i]; acc += x * x; }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[sg * kKerasSlices + sl] = (float)sh[0];
}

// grid.y = tensor: every block first derives its tensor's clip factor from the 16 partials (same order everywhere)
__global__ __launch_bounds__(256) void adam_keras_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, const KerasSegs segs,
                                                         const float* __restrict__ partial, float clipnorm, float lr_t,
                                                         float beta1, float beta2, float eps) {
  const int sg = blockIdx.y;
  double sq = 0.0;
#pragma unroll
  for (int j = 0; j < kKerasSlices; ++j) sq += (double)partial[sg * kKerasSlices + j];
  const float norm = (float)sqrt(sq);
  const float scale = (clipnorm > 0.f && norm > clipnorm) ? clipnorm / norm : 1.f;
  const long long n = segs.size[sg], base = segs.off[sg];
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long e = base + i;
    const float gg = g[e] * scale;
    float mm = m[e], vv = v[e];
    mm += (gg - mm) * omb1;
    vv += (gg * gg - vv) * omb2;
    m[e] = mm; v[e] = vv;
    p[e] -= lr_t * mm / (sqrtf(vv) + eps);
  }
}

int launch_adam_keras(float* param, const float* grad, float* m, float* v, int n_seg, const int64_t* seg_off,
                      const int64_t* seg_size, float clipnorm, float lr_t, float beta1, float beta2, float eps,
                      float* scratch, hipStream_t st) {
  XT_REQUIRE(param && grad && m && v && seg_off && seg_size && scratch, "xt_adam_keras: null argument");
  XT_REQUIRE(n_seg > 0 && n_seg <= kKerasMaxSeg, "xt_adam_keras: n_seg %d outside (0,%d]", n_seg, kKerasMaxSeg);
  KerasSegs segs;
  segs.n = n_seg;
  long long biggest = 1;
  for (int i = 0; i < n_seg; ++i) {
    XT_REQUIRE(seg_off[i] >= 0 && seg_size[i] > 0, "xt_adam_keras: bad segment %d", i);
    segs.off[i] = seg_off[i]; segs.size[i] = seg_size[i];
    if (seg_size[i] > biggest) biggest = seg_size[i];
  }
  hipLaunchKernelGGL(keras_seg_sqnorm_kernel, dim3(kKerasSlices, n_seg), dim3(256), 0, st, grad, segs, scratch);
  XT_LAUNCH_CHECK();
  int gx = (int)((biggest + 255) / 256);
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(adam_keras_kernel, dim3(gx, n_seg), dim3(256), 0, st, param, grad, m, v, segs, scratch, clipnorm, lr_t,
                     beta1, beta2, eps);
  XT_LAUNCH_CHECK();
  return 0;
}

int launch_norm_finalize(const float* partial, int nblocks, float clip_norm, float grad_scale, float lr, float beta1,
                         float beta2, int advance, float* state, const LossArgs* la, hipStream_t st,
                         const float* lr_dev = nullptr) {
  LossArgs l{};
  if (la) l = *la;
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nblocks, clip_norm, grad_scale, lr, beta1,
                     beta2, advance, state, l, lr_dev);
  XT_LAUNCH_CHECK();
  return 0;
}

// only the per-block squared-norm partials of a flat gradient (the first launch of launch_global_norm): for a consumer that
// derives the clip factor itself (adam_tf_clip_kernel) -- the data-parallel step after the exchange
int launch_sqnorm_partial(const float* grad, long long count, float* scratch, int* nblocks_out, hipStream_t st) {
  XT_REQUIRE(count > 0 && grad && scratch && nblocks_out, "sqnorm_partial: bad arguments");
  XT_REQUIRE(((uintptr_t)grad & 15) == 0, "sqnorm_partial: grad must be 16-byte aligned");
  int nb = (int)((count / 4 + 255) / 256);
  if (nb > kNormBlocks) nb = kNormBlocks;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(256), 0, st, grad, count, scratch);
  XT_LAUNCH_CHECK();
  *nblocks_out = nb;
  return 0;
}

int launch_global_norm(const float* grad, long long count, float clip_norm, float grad_scale, float lr, float beta1,
                       float beta2, int advance, float* state, float* scratch, hipStream_t st, const float* lr_dev,
                       int* nblocks_out) {
  XT_REQUIRE(count > 0 && grad && state && scratch, "global_norm: bad arguments");
  XT_REQUIRE(((uintptr_t)grad & 15) == 0, "global_norm: grad must be 16-byte aligned");
  int nb = (int)((count / 4 + 255) / 256);
  if (nb > kNormBlocks) nb = kNormBlocks;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(256), 0, st, grad, count, scratch);
  XT_LAUNCH_CHECK();
  if (nblocks_out) *nblocks_out = nb;
  return launch_norm_finalize(scratch, nb, clip_norm, grad_scale, lr, beta1, beta2, advance, state, nullptr, st, lr_dev);
}

// ---- stand-alone forms of the data-parallel tail for the steps whose gradient reduction / optimiser launches do not carry
// it: the overlapped two-bucket exchange (the tail must be in the FIRST bucket, before the backward has finished) and a rank
// whose trajectory shard of a chunk is empty (no gradient-reduction launch at all: zeros, the step-size advance, the tail)
__global__ void dp_tail_write_kernel(float* __restrict__ tail, int rank, float rows, const float* __restrict__ loss,
                                     float* __restrict__ state, float lr, const float* __restrict__ lr_dev, float beta1,
                                     float beta2, int advance) {
  if (threadIdx.x == 0 && advance) adam_advance(state, lr_dev ? lr_dev[0] : lr, beta1, beta2);
  if (threadIdx.x < kDpTailFloats) tail[threadIdx.x] = dp_tail_value(threadIdx.x, rank, rows, loss ? loss[0] : 0.f);
}
__global__ void dp_tail_consume_kernel(const DpStep dp) { dp_tail_consume(dp, true); }

int launch_dp_tail_write(float* tail, int rank, float rows, const float* loss, float* state, float lr, const float* lr_dev,
                         float beta1, float beta2, int advance, hipStream_t st) {
  XT_REQUIRE(tail && rank >= 0 && rank < kDpMaxWorld, "dp_tail_write: bad arguments");
  hipLaunchKernelGGL(dp_tail_write_kernel, dim3(1), dim3(64), 0, st, tail, rank, rows, loss, state, lr, lr_dev, beta1, beta2,
                     advance);
  XT_LAUNCH_CHECK();
  return 0;
}
int launch_dp_tail_consume(const DpStep* dp, hipStream_t st) {
  XT_REQUIRE(dp && dp->tail && !dp->flags, "dp_tail_consume: bad arguments");
  hipLaunchKernelGGL(dp_tail_consume_kernel, dim3(1), dim3(64), 0, st, *dp);
  XT_LAUNCH_CHECK();
  return 0;
}

int launch_adam(float* param, const float* grad, float* m, float* v, long long count, float beta1, float beta2,
                float eps, const float* state, hipStream_t st) {
  XT_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
             "adam: buffers must be 16-byte aligned");
  int nb = (int)((count / 4 + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(adam_tf_kernel, dim3(nb), dim3(256), 0, st, param, grad, m, v, count, beta1, beta2, eps, state);
  XT_LAUNCH_CHECK();
  return 0;
}

}  // namespace xt

extern "C" {

int xt_adam_state_init(float* state, void* stream) {
  hipLaunchKernelGGL(xt::adam_state_init_kernel, dim3(1), dim3(64), 0, xt::as_stream(stream), state);
  XT_LAUNCH_CHECK();
  return 0;
}

int xt_grad_global_norm(const float* grad, int64_t count, float clip_norm, float grad_scale, float* state,
                        float* scratch, void* stream) {
  return xt::launch_global_norm(grad, count, clip_norm, grad_scale, 0.f, 0.f, 0.f, 0, state, scratch,
                                xt::as_stream(stream), nullptr, nullptr);
}

int xt_adam_keras(float* param, const float* grad, float* m, float* v, int32_t n_seg, const int64_t* seg_off,
                  const int64_t* seg_size, float clipnorm, float lr_t, float beta1, float beta2, float eps,
                  float* scratch, void* stream) {
  return xt::launch_adam_keras(param, grad, m, v, n_seg, seg_off, seg_size, clipnorm, lr_t, beta1, beta2, eps, scratch,
                               xt::as_stream(stream));
}

int xt_adam_tf_clip(float* param, const float* grad, float* m, float* v, int64_t count, float lr, float beta1,
                    float beta2, float eps, float clip_norm, float grad_scale, float* state, float* scratch,
                    void* stream) {
  if (int rc = xt::launch_global_norm(grad, count, clip_norm, grad_scale, lr, beta1, beta2, 1, state, scratch,
                                      xt::as_stream(stream), nullptr, nullptr))
    return rc;
  return xt::launch_adam(param, grad, m, v, count, beta1, beta2, eps, state, xt::as_stream(stream));
}

}  // extern "C"
