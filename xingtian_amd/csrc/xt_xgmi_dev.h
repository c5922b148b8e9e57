// Device-side vocabulary of the direct 2-phase gradient exchange (xt_xgmi.hip) shared with the kernels that take part in
// it when the exchange is fused into the SGD step (xt_optim.hip: grads_finish_kernel pushes its output straight into the
// owners' inboxes, the optimiser kernels read the reduced gradient out of the exchange block).  The reference's only
// counterpart is the host-side float64 sum through a RawArray, xt/framework/trainer.py:86-92,139-144 (dead code there).
#pragma once
#include "xt_common.h"

namespace xt {

constexpr int kMaxWorld = kDpMaxWorld;
constexpr int kFlagWords = 1024;            // uint32 words at the head of the exchange block: ready[64], done[64], identity
constexpr int kReadyOff = 0, kDoneOff = 64;
constexpr int kIdentOff = 512;              // [512..516): 16 bytes identifying the DEVICE the block lives on (hipDeviceProp uuid)
// device-local control words (one small hipMalloc, NOT shared): [0] seq of the last completed all-reduce, [1] error bits,
// [2] reduce ticket, [3] gather ticket, [8 + q] scatter tickets per peer
constexpr int kCtlSeq = 0, kCtlErr = 1, kCtlRed = 2, kCtlGat = 3, kCtlScat = 8, kCtlWords = 8 + kMaxWorld;
// error bits of the control word (xt_direct_status; sticky)
constexpr uint32_t kErrScatterWait = 1u, kErrReduceWait = 2u, kErrRowsMismatch = 4u;

__device__ __forceinline__ void slice_of(int64_t nvec, int r, int world, int64_t& b, int64_t& e) {
  const int64_t base = nvec / world, rem = nvec % world;
  b = r * base + (r < rem ? r : rem);
  e = b + base + (r < rem ? 1 : 0);
}

// owner of float4 `v` under the balanced contiguous split of nvec over `world` slices (the first nvec % world slices hold
// one vector more), and the first vector of that owner's slice
__device__ __forceinline__ int owner_of(int64_t v, int64_t nvec, int world, int64_t& slice_begin) {
  const int64_t base = nvec / world, rem = nvec % world;
  const int64_t cut = rem * (base + 1);
  const int q = v < cut ? (int)(v / (base + 1)) : (int)(rem + (base ? (v - cut) / base : 0));
  slice_begin = q * base + (q < rem ? q : rem);
  return q;
}

// Publishing a block's share of the data.  The exchange block is UNCACHED device memory: stores bypass every cache, so a
// block only has to wait until its own stores are acknowledged (s_waitcnt vmcnt(0)) before it takes its ticket, and a
// consumer has nothing to invalidate.  (First version: a system-scope release fence per block -- a buffer_wbl2 of the XCD's
// L2, needed per block because eight XCDs have eight L2s -- and an acquire fence per consumer block: 100 / 155 / 303 us for
// 2 / 4 / 8 in-process ranks x 3.39 MB against 57 / 74 / 142 us without them, tools/direct_probe.py.)
__device__ __forceinline__ void publish_fence() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// bounded wait until flags[off + p] == seq for every p < world (thread 0 of the block polls, the block follows)
__device__ __forceinline__ void wait_all(const uint32_t* flags, int off, int world, uint32_t seq, uint32_t* ctl,
                                         unsigned long long timeout_ticks, uint32_t errbit) {
  // the error word is sticky: once a wait of this comm has run out, later waits do not wait again (a comm that lost a peer
  // costs ONE time-out, then every kernel runs through and xt_direct_status / the loss read-back tell the host)
  if (threadIdx.x == 0 && __hip_atomic_load(ctl + kCtlErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
    const unsigned long long t0 = wall_clock64();
    for (int p = 0; p < world; ++p) {
      int spins = 0;
      while (__hip_atomic_load(flags + off + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
        __builtin_amdgcn_s_sleep(4);
        if ((++spins & 63) == 0 && wall_clock64() - t0 > timeout_ticks) {
          atomicOr(ctl + kCtlErr, errbit);
          p = world;
          break;
        }
      }
    }
  }
  __syncthreads();
  asm volatile("" ::: "memory");            // (uncached data: nothing to invalidate; keep the loads below the wait)
}

}  // namespace xt
