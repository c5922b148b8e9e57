// Direct 2-phase all-reduce of the flat fp32 gradient over peer-mapped memory (SURVEY.md section 8(b) `xt_allreduce_direct`,
// section 8(e): "RCCL ncclAllReduce first; then the 2-phase direct xGMI all-reduce because ring is per-link bound").
//
// The reference's only multi-device exchange is the dead host-side trainer (xt/framework/trainer.py:86-92,139-144): every
// process writes its whole flat parameter-gradient list into a shared RawArray, the sum is taken in float64 on the host.
// Here: one process per GPU, every rank owns ONE device allocation ("exchange block": flags + inbox + result) that all
// peers map through hipIpcGetMemHandle / hipIpcOpenMemHandle; an all-reduce is three small kernels on the caller's stream
//
//   scatter   every rank PUSHES slice q of its gradient into peer q's inbox[rank]  (posted remote writes, no remote reads),
//             then raises ready_q[rank] = seq
//   reduce    rank r waits for ready_r[*] == seq, sums its slice over inbox[0..N-1] in FIXED rank order (every element of
//             the result is computed by exactly one rank -> all replicas receive bit-identical values) and PUSHES the
//             reduced slice into every peer's result buffer, then raises done_p[r] = seq on every peer
//   gather    rank r waits for done_r[*] == seq and copies result -> gradient buffer
//
// xGMI is a point-to-point mesh (7 links per GPU): both phases talk to all N-1 peers at once, 2 (N-1)/N S bytes leave every
// GPU spread over N-1 links, against a ring's 2 (N-1)/N S bytes through ONE link pair in 2 (N-1) dependent hops.
// The sequence number lives in device memory and is advanced by the gather kernel, so the three launches can be captured
// into the hipGraph of a whole update (xt_net_ppo_train) and replayed.
//
// Memory model: the exchange block is allocated uncached / fine-grained (hipExtMallocWithFlags), data is published with a
// system-scope release (fence + atomic store of the flag) and consumed behind a system-scope acquire -- coarse-grained
// memory would only be coherent between devices at kernel boundaries.  Every wait is BOUNDED (timeout -> error word, the
// kernels run to completion with whatever arrived): a lost peer cannot hang the GPU.
//
// Verified on one device only (N processes or N in-process ranks sharing the GPU, tests/test_gpu_direct.py): no multi-GPU
// box was available to the builders; the kernels follow the HIP memory model for peer access, timing over xGMI is open.
#include <string.h>

#include "xt_xgmi_dev.h"

namespace xt {

constexpr int kScatterVecs = 1024;          // float4 per block in scatter / gather (16 KB)
constexpr int kReduceVecs = 512;            // float4 per block in reduce

__device__ __forceinline__ float4 load_vec_tail(const float* p, int64_t v, int64_t count) {
  // vec v of a buffer of `count` floats (the last vec may be partial: zero filled)
  const int64_t i = v * 4;
  if (i + 4 <= count) return *reinterpret_cast<const float4*>(p + i);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < count) r.x = p[i];
  if (i + 1 < count) r.y = p[i + 1];
  if (i + 2 < count) r.z = p[i + 2];
  return r;
}

// phase 1: push slice q of the local gradient into peer q's inbox[rank]; grid (blocks per slice, world)
__global__ void __launch_bounds__(256) xgmi_scatter_kernel(const float* __restrict__ grads, int64_t count, int rank, int world,
                                                           DirectPeers peers, uint32_t* ctl) {
  const int q = blockIdx.y;
  const int64_t nvec = (count + 3) / 4;
  int64_t b, e;
  slice_of(nvec, q, world, b, e);
  const int64_t lo = b + (int64_t)blockIdx.x * kScatterVecs;
  const int64_t hi = lo + kScatterVecs < e ? lo + kScatterVecs : e;
  float4* dst = reinterpret_cast<float4*>(peers.inbox_me[q]);
  for (int64_t v = lo + threadIdx.x; v < hi; v += 256) dst[v - b] = load_vec_tail(grads, v, count);
  publish_fence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t seq = ctl[kCtlSeq] + 1;
    if (atomicAdd(ctl + kCtlScat + q, 1u) == gridDim.x - 1) {     // last block of this peer's slice
      __hip_atomic_store(peers.flags[q] + kReadyOff + rank, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// phase 2: sum my slice over the inbox slots in rank order, push the result to every peer; grid = blocks over my slice
__global__ void __launch_bounds__(256) xgmi_reduce_kernel(const float* __restrict__ inbox, int64_t slice_cap, int64_t count,
                                                          int rank, int world, DirectPeers peers, const uint32_t* my_flags,
                                                          uint32_t* ctl, unsigned long long timeout_ticks) {
  const uint32_t seq = ctl[kCtlSeq] + 1;
  wait_all(my_flags, kReadyOff, world, seq, ctl, timeout_ticks, 1u);
  const int64_t nvec = (count + 3) / 4;
  int64_t b, e;
  slice_of(nvec, rank, world, b, e);
  const int64_t lo = b + (int64_t)blockIdx.x * kReduceVecs;
  const int64_t hi = lo + kReduceVecs < e ? lo + kReduceVecs : e;
  for (int64_t v = lo + threadIdx.x; v < hi; v += 256) {
    float4 acc = *reinterpret_cast<const float4*>(inbox + (v - b) * 4);
    for (int p = 1; p < world; ++p) {           // FIXED order 0, 1, ..., N-1: the sum is a function of the data only
      const float4 x = *reinterpret_cast<const float4*>(inbox + p * slice_cap + (v - b) * 4);
      acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
    }
    for (int p = 0; p < world; ++p) reinterpret_cast<float4*>(peers.result[p])[v] = acc;
  }
  publish_fence();
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(ctl + kCtlRed, 1u) == gridDim.x - 1) {
    for (int p = 0; p < world; ++p)
      __hip_atomic_store(peers.flags[p] + kDoneOff + rank, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// phase 3: wait for every rank's reduced slice, copy result -> gradient buffer; the last block advances the sequence
__global__ void __launch_bounds__(256) xgmi_gather_kernel(float* __restrict__ grads, const float* __restrict__ result,
                                                          int64_t count, int world, const uint32_t* my_flags, uint32_t* ctl,
                                                          unsigned long long timeout_ticks) {
  const uint32_t seq = ctl[kCtlSeq] + 1;
  wait_all(my_flags, kDoneOff, world, seq, ctl, timeout_ticks, 2u);
  const int64_t nvec = (count + 3) / 4;
  const int64_t lo = (int64_t)blockIdx.x * kScatterVecs;
  const int64_t hi = lo + kScatterVecs < nvec ? lo + kScatterVecs : nvec;
  for (int64_t v = lo + threadIdx.x; v < hi; v += 256) {
    const float4 x = reinterpret_cast<const float4*>(result)[v];
    const int64_t i = v * 4;
    if (i + 4 <= count) {
      *reinterpret_cast<float4*>(grads + i) = x;
    } else {
      if (i < count) grads[i] = x.x;
      if (i + 1 < count) grads[i + 1] = x.y;
      if (i + 2 < count) grads[i + 2] = x.z;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(ctl + kCtlGat, 1u) == gridDim.x - 1) {
      ctl[kCtlRed] = 0;
      ctl[kCtlGat] = 0;
      for (int q = 0; q < world; ++q) ctl[kCtlScat + q] = 0;
      __threadfence();
      ctl[kCtlSeq] = seq;
    }
  }
}

// The three phases in ONE launch (per-process path): a block pushes its chunk of the gradient (scatter), then -- without
// leaving the kernel -- waits for the peers' chunks of ITS share of this rank's slice, reduces and pushes it (reduce), then
// waits for the reduced slices and copies its chunk of the result back (gather).  Cross-RANK dependencies only (flags); no
// block ever waits for another block of its own launch, so the grid needs no barrier -- but every block of every rank's
// launch must be resident, which 104-208 blocks per rank are, also with eight ranks on one GPU.  Saves two kernel
// boundaries and two cold prologues per all-reduce (tools/direct_probe.py --procs: see DESIGN.md section 5).
// Tickets count VECTORS (a block's scatter chunk may straddle two slices): whoever completes a slice raises its flag.
constexpr int kFusedVecs = 2048;            // float4 per block in the scatter / gather phases (32 KB)
__global__ void __launch_bounds__(256) xgmi_fused_kernel(float* __restrict__ grads, const float* __restrict__ inbox,
                                                         const float* __restrict__ result, int64_t slice_cap, int64_t count,
                                                         int rank, int world, DirectPeers peers, const uint32_t* my_flags,
                                                         uint32_t* ctl, unsigned long long timeout_ticks) {
  const uint32_t seq = ctl[kCtlSeq] + 1;
  const int64_t nvec = (count + 3) / 4;
  __shared__ uint32_t s_cnt[kMaxWorld];
  if (threadIdx.x < kMaxWorld) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  // ---- scatter (the grid is CAPPED at what can be resident next to the other ranks' kernels: chunks are grid-strided)
  {
    for (int64_t lo = (int64_t)blockIdx.x * kFusedVecs; lo < nvec; lo += (int64_t)gridDim.x * kFusedVecs) {
      const int64_t hi = lo + kFusedVecs < nvec ? lo + kFusedVecs : nvec;
      for (int64_t v = lo + threadIdx.x; v < hi; v += 256) {
        int64_t b;
        const int q = owner_of(v, nvec, world, b);
        reinterpret_cast<float4*>(peers.inbox_me[q])[v - b] = load_vec_tail(grads, v, count);
        atomicAdd(&s_cnt[q], 1u);
      }
    }
    publish_fence();
    __syncthreads();
    if (threadIdx.x < world) {
      const int q = threadIdx.x;
      int64_t b, e;
      slice_of(nvec, q, world, b, e);
      const uint32_t mine = s_cnt[q];
      // (an EMPTY slice -- fewer vectors than ranks -- has no last contributor: block 0 raises its flag)
      const bool last = mine ? atomicAdd(ctl + kCtlScat + q, mine) + mine == (uint32_t)(e - b) : (e == b && blockIdx.x == 0);
      if (last) __hip_atomic_store(peers.flags[q] + kReadyOff + rank, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // ---- reduce: this block's share of my slice
  {
    wait_all(my_flags, kReadyOff, world, seq, ctl, timeout_ticks, 1u);
    int64_t b, e;
    slice_of(nvec, rank, world, b, e);
    const int64_t per = ((e - b) + gridDim.x - 1) / gridDim.x;
    const int64_t lo = b + (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < e ? lo + per : e;
    for (int64_t v = lo + threadIdx.x; v < hi; v += 256) {
      float4 acc = *reinterpret_cast<const float4*>(inbox + (v - b) * 4);
      for (int p = 1; p < world; ++p) {           // FIXED order 0, 1, ..., N-1
        const float4 x = *reinterpret_cast<const float4*>(inbox + p * slice_cap + (v - b) * 4);
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
      }
      for (int p = 0; p < world; ++p) reinterpret_cast<float4*>(peers.result[p])[v] = acc;
    }
    publish_fence();
    __syncthreads();
    const uint32_t mine = hi > lo ? (uint32_t)(hi - lo) : 0u;
    if (threadIdx.x == 0) {
      const bool last = mine ? atomicAdd(ctl + kCtlRed, mine) + mine == (uint32_t)(e - b) : (e == b && blockIdx.x == 0);
      if (last)
        for (int p = 0; p < world; ++p)
          __hip_atomic_store(peers.flags[p] + kDoneOff + rank, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // ---- gather
  wait_all(my_flags, kDoneOff, world, seq, ctl, timeout_ticks, 2u);
  for (int64_t lo = (int64_t)blockIdx.x * kFusedVecs; lo < nvec; lo += (int64_t)gridDim.x * kFusedVecs) {
    const int64_t hi = lo + kFusedVecs < nvec ? lo + kFusedVecs : nvec;
    for (int64_t v = lo + threadIdx.x; v < hi; v += 256) {
      const float4 x = reinterpret_cast<const float4*>(result)[v];
      const int64_t i = v * 4;
      if (i + 4 <= count) {
        *reinterpret_cast<float4*>(grads + i) = x;
      } else {
        if (i < count) grads[i] = x.x;
        if (i + 1 < count) grads[i + 1] = x.y;
        if (i + 2 < count) grads[i + 2] = x.z;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(ctl + kCtlGat, 1u) == gridDim.x - 1) {
      ctl[kCtlRed] = 0;
      ctl[kCtlGat] = 0;
      for (int q = 0; q < world; ++q) ctl[kCtlScat + q] = 0;
      __threadfence();
      ctl[kCtlSeq] = seq;
    }
  }
}

}  // namespace xt

struct xt_direct_comm {
  int rank = 0, world = 1;
  int64_t max_count = 0, slice_cap = 0;         // floats
  size_t block_bytes = 0, norm_off = 0, inbox_off = 0, result_off = 0;
  char* block = nullptr;                        // own exchange block (device memory, shared with the peers)
  char* peer_block[xt::kMaxWorld] = {};
  bool peer_ipc[xt::kMaxWorld] = {};            // opened with hipIpcOpenMemHandle (to be closed)
  bool connected = false;
  uint32_t* ctl = nullptr;                      // device-local control words
  unsigned long long timeout_ticks = 200000000ull;   // 2 s of the 100 MHz wall clock
  int calls = 0;
  int fused = 1;                                // xt_allreduce_direct: one launch (xgmi_fused_kernel) instead of three
  int ranks_on_device = 1;                      // ranks of the group whose exchange block lives on THIS device
  int resident_blocks = 0;                      // 256-thread workgroups the device can hold at once
  xt::DirectPeers peers = {};
};

namespace xt {

// workgroups a kernel of this comm whose blocks SPIN on other ranks' flags may be launched with: all of them must be
// resident together with the same kernel of every other rank that shares the device (N test processes on one GPU) --
// otherwise producer blocks queue behind spinning ones until the bounded wait runs out
static int spin_block_cap(const xt_direct_comm* c) {
  const int res = c->resident_blocks > 0 ? c->resident_blocks : 2048;
  const int cap = res / (c->ranks_on_device > 0 ? c->ranks_on_device : 1);
  return cap < 1 ? 1 : cap;
}

static int reduce_blocks(int64_t nvec, int world) {
  const int64_t slice_max = (nvec + world - 1) / world;
  int64_t rb = (slice_max + 255) / 256;
  if (rb > kDpRedBlocksMax) rb = kDpRedBlocksMax;
  return rb < 1 ? 1 : (int)rb;
}

// ---- the pieces of the exchange fused into the SGD step (called by xt_net.hip)
int direct_fill_finish(xt_direct_comm* c, int64_t count, DpFinish* f) {
  XT_REQUIRE(c && f && c->connected, "direct exchange: the comm is not connected");
  XT_REQUIRE(count > 0 && count <= c->max_count && count % 4 == 0, "direct exchange: count %lld outside (0, %lld] or not a multiple of 4",
             (long long)count, (long long)c->max_count);
  XT_REQUIRE(count / 4 >= c->world, "direct exchange: %lld float4s cannot be split over %d ranks", (long long)(count / 4), c->world);
  f->scatter = 1; f->rank = c->rank; f->world = c->world; f->nvec = count / 4; f->peers = c->peers; f->ctl = c->ctl;
  return 0;
}

int direct_launch_scatter(xt_direct_comm* c, const float* buf, int64_t count, hipStream_t st);

// what the optimiser kernel of a fused step needs: the inbox to reduce (its first `red_blocks` workgroups sum this rank's
// slice in rank order and push it, with its squared-norm partials, to every peer), the flags to wait for, the reduced buffer
// and all ranks' partials to read.  red_blocks depends on (count, world) only: the same on every rank.
int direct_fill_step(xt_direct_comm* c, int64_t count, int64_t count_grad, DpStep* s, const float** result,
                     const float** partial, int* npartial, int* block_cap) {
  XT_REQUIRE(c && s && c->connected, "direct exchange: the comm is not connected");
  XT_REQUIRE(count > 0 && count <= c->max_count, "direct exchange: count %lld outside (0, %lld]", (long long)count,
             (long long)c->max_count);
  const int64_t nvec = (count + 3) / 4;
  const int rb = reduce_blocks(nvec, c->world);
  XT_REQUIRE(spin_block_cap(c) >= rb, "direct exchange: %d reduce workgroups cannot be resident with %d ranks on this device",
             rb, c->ranks_on_device);
  s->flags = reinterpret_cast<const uint32_t*>(c->block);
  s->ctl = c->ctl;
  s->timeout_ticks = c->timeout_ticks;
  s->world = c->world;
  s->inbox = reinterpret_cast<const float*>(c->block + c->inbox_off);
  s->slice_cap = c->slice_cap; s->nvec = nvec; s->nvec_grad = (count_grad + 3) / 4;
  s->rank = c->rank; s->red_blocks = rb; s->peers = c->peers;
  *result = reinterpret_cast<const float*>(c->block + c->result_off);
  *partial = reinterpret_cast<const float*>(c->block + c->norm_off);
  *npartial = c->world * rb;
  *block_cap = spin_block_cap(c);
  c->calls++;
  return 0;
}

}  // namespace xt

extern "C" {

int xt_direct_create(int32_t rank, int32_t world, int64_t max_count, void* handle_out, xt_direct_comm** out) {
  XT_REQUIRE(out, "xt_direct_create: null out");
  XT_REQUIRE(world >= 1 && world <= xt::kMaxWorld && rank >= 0 && rank < world, "xt_direct_create: rank %d / world %d (max %d)",
             rank, world, xt::kMaxWorld);
  XT_REQUIRE(max_count > 0 && max_count < (1ll << 31), "xt_direct_create: max_count %lld", (long long)max_count);
  xt_direct_comm* c = new xt_direct_comm();
  c->rank = rank; c->world = world; c->max_count = max_count;
  const int64_t nvec = (max_count + 3) / 4;
  c->slice_cap = ((nvec + world - 1) / world) * 4;
  c->norm_off = (size_t)xt::kFlagWords * 4;
  c->inbox_off = c->norm_off + sizeof(float) * (size_t)xt::kMaxWorld * xt::kDpRedBlocksMax;
  c->result_off = c->inbox_off + sizeof(float) * (size_t)c->slice_cap * world;
  c->block_bytes = c->result_off + sizeof(float) * (size_t)nvec * 4;
  // UNCACHED device memory (what RCCL takes for its peer buffers) and nothing else.  Measured round 5 with every rank on ONE
  // device: plain hipMalloc memory -> sporadic 5 s time-outs of the flag polls (stale lines served by another XCD's L2:
  // coarse-grained memory is only coherent at kernel boundaries); fine-grained memory with system-scope fences per block ->
  // no time-outs but WRONG sums (and 2.5x the time).  The uncached form has passed every run of tests/test_gpu_direct.py.
  hipError_t e = hipExtMallocWithFlags((void**)&c->block, c->block_bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    xt::set_error("xt_direct_create: cannot allocate the %zu-byte uncached exchange block: %s", c->block_bytes, hipGetErrorString(e));
    delete c;
    return 1;
  }
  XT_CHECK_HIP(hipMemset(c->block, 0, c->block_bytes));
  XT_CHECK_HIP(hipMalloc((void**)&c->ctl, sizeof(uint32_t) * xt::kCtlWords));
  XT_CHECK_HIP(hipMemset(c->ctl, 0, sizeof(uint32_t) * xt::kCtlWords));
  {
    // the identity of the DEVICE this block lives on, left in the block for the peers: lets every rank count how many ranks
    // share its device (xt_direct_info) and size its spinning launches for that
    int dev = 0, cus = 0, per_cu = 0;
    XT_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    XT_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    uint32_t ident[4] = {0, 0, 0, 0};
    memcpy(ident, prop.uuid.bytes, sizeof(ident));
    if ((ident[0] | ident[1] | ident[2] | ident[3]) == 0u) {      // (no uuid: PCI location)
      ident[0] = 0x58544456u; ident[1] = (uint32_t)prop.pciDomainID; ident[2] = (uint32_t)prop.pciBusID; ident[3] = (uint32_t)prop.pciDeviceID;
    }
    XT_CHECK_HIP(hipMemcpy(c->block + (size_t)xt::kIdentOff * 4, ident, sizeof(ident), hipMemcpyHostToDevice));
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, xt::xgmi_fused_kernel, 256, 0) == hipSuccess)
      c->resident_blocks = cus * per_cu;
    (void)hipGetLastError();
  }
  XT_CHECK_HIP(hipDeviceSynchronize());
  if (handle_out) {
    hipIpcMemHandle_t h;
    static_assert(sizeof(hipIpcMemHandle_t) == XT_DIRECT_HANDLE_BYTES, "hipIpcMemHandle_t size");
    e = hipIpcGetMemHandle(&h, c->block);
    if (e != hipSuccess && world > 1) {
      (void)hipGetLastError();
      xt::set_error("xt_direct_create: hipIpcGetMemHandle failed: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)", hipGetErrorString(e));
      (void)hipFree(c->block); (void)hipFree(c->ctl); delete c;
      return 1;
    }
    if (e != hipSuccess) { (void)hipGetLastError(); memset(&h, 0, sizeof(h)); }
    memcpy(handle_out, &h, sizeof(h));
  }
  *out = c;
  return 0;
}

static int direct_finish_connect(xt_direct_comm* c) {
  uint32_t mine[4] = {0, 0, 0, 0};
  XT_CHECK_HIP(hipMemcpy(mine, c->block + (size_t)xt::kIdentOff * 4, sizeof(mine), hipMemcpyDeviceToHost));
  int same = 0;
  for (int q = 0; q < c->world; ++q) {
    char* blk = c->peer_block[q];
    c->peers.flags[q] = reinterpret_cast<uint32_t*>(blk);
    c->peers.norm_part[q] = reinterpret_cast<float*>(blk + c->norm_off);
    c->peers.inbox_me[q] = reinterpret_cast<float*>(blk + c->inbox_off) + (size_t)c->rank * c->slice_cap;
    c->peers.result[q] = reinterpret_cast<float*>(blk + c->result_off);
    uint32_t theirs[4] = {1, 1, 1, 1};
    if (q == c->rank || hipMemcpy(theirs, blk + (size_t)xt::kIdentOff * 4, sizeof(theirs), hipMemcpyDeviceToHost) == hipSuccess) {
      if (q == c->rank || memcmp(mine, theirs, sizeof(mine)) == 0) ++same;
    } else {
      (void)hipGetLastError();
    }
  }
  c->ranks_on_device = same < 1 ? 1 : same;
  c->connected = true;
  return 0;
}

int xt_direct_connect(xt_direct_comm* c, const void* handles) {
  XT_REQUIRE(c && (handles || c->world == 1), "xt_direct_connect: null argument");
  XT_REQUIRE(!c->connected, "xt_direct_connect: already connected");
  for (int q = 0; q < c->world; ++q) {
    if (q == c->rank) { c->peer_block[q] = c->block; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, static_cast<const char*>(handles) + (size_t)q * XT_DIRECT_HANDLE_BYTES, sizeof(h));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      xt::set_error("xt_direct_connect: hipIpcOpenMemHandle of rank %d's exchange block failed: %s", q, hipGetErrorString(e));
      return 1;
    }
    c->peer_block[q] = static_cast<char*>(p);
    c->peer_ipc[q] = true;
  }
  return direct_finish_connect(c);
}

int xt_direct_connect_local(xt_direct_comm* c, xt_direct_comm* const* ranks) {
  XT_REQUIRE(c && ranks, "xt_direct_connect_local: null argument");
  XT_REQUIRE(!c->connected, "xt_direct_connect_local: already connected");
  for (int q = 0; q < c->world; ++q) {
    XT_REQUIRE(ranks[q] && ranks[q]->rank == q && ranks[q]->world == c->world && ranks[q]->max_count == c->max_count,
               "xt_direct_connect_local: entry %d is not rank %d of the same group", q, q);
    c->peer_block[q] = ranks[q]->block;
  }
  return direct_finish_connect(c);
}

// phase 0 = scatter, 1 = reduce, 2 = gather, -1 = all three
static int direct_enqueue(xt_direct_comm* c, float* buf, int64_t count, int phase, hipStream_t st) {
  XT_REQUIRE(c && buf, "xt_allreduce_direct: null argument");
  XT_REQUIRE(count > 0 && count <= c->max_count, "xt_allreduce_direct: count %lld outside (0, %lld]", (long long)count,
             (long long)c->max_count);
  XT_REQUIRE((reinterpret_cast<uintptr_t>(buf) & 15) == 0, "xt_allreduce_direct: the buffer must be 16-byte aligned");
  if (phase <= 0) c->calls++;
  if (c->world == 1) return 0;
  XT_REQUIRE(c->connected, "xt_allreduce_direct: xt_direct_connect has not been called");
  const int64_t nvec = (count + 3) / 4;
  const int64_t slice_max = (nvec + c->world - 1) / c->world;
  const unsigned sb = (unsigned)((slice_max + xt::kScatterVecs - 1) / xt::kScatterVecs);
  const unsigned rb = (unsigned)((slice_max + xt::kReduceVecs - 1) / xt::kReduceVecs);
  const unsigned gb = (unsigned)((nvec + xt::kScatterVecs - 1) / xt::kScatterVecs);
  // the reduce / gather blocks of the three-launch form spin on other ranks' flags: all of them must fit next to the
  // other ranks' launches (the scatter blocks never wait).  One block covers a fixed chunk, so a buffer too large for the cap
  // is refused instead of risking the bounded wait (the fused single launch grid-strides and has no such limit)
  XT_REQUIRE((int)rb <= xt::spin_block_cap(c) && (int)gb <= xt::spin_block_cap(c),
             "xt_allreduce_direct (three launches): %u reduce / %u gather workgroups exceed the %d that can be resident with %d "
             "rank(s) on this device; use the fused form (xt_direct_set_fused)", rb, gb, xt::spin_block_cap(c), c->ranks_on_device);
  const uint32_t* my_flags = reinterpret_cast<const uint32_t*>(c->block);
  const float* inbox = reinterpret_cast<const float*>(c->block + c->inbox_off);
  const float* result = reinterpret_cast<const float*>(c->block + c->result_off);
  if (phase < 0 || phase == 0)
    hipLaunchKernelGGL(xt::xgmi_scatter_kernel, dim3(sb ? sb : 1, c->world), dim3(256), 0, st, buf, count, c->rank, c->world,
                       c->peers, c->ctl);
  if (phase < 0 || phase == 1)
    hipLaunchKernelGGL(xt::xgmi_reduce_kernel, dim3(rb ? rb : 1), dim3(256), 0, st, inbox, c->slice_cap, count, c->rank,
                       c->world, c->peers, my_flags, c->ctl, c->timeout_ticks);
  if (phase < 0 || phase == 2)
    hipLaunchKernelGGL(xt::xgmi_gather_kernel, dim3(gb ? gb : 1), dim3(256), 0, st, buf, result, count, c->world, my_flags,
                       c->ctl, c->timeout_ticks);
  XT_LAUNCH_CHECK();
  return 0;
}

int xt_allreduce_direct(xt_direct_comm* c, float* buf, int64_t count, void* stream) {
  if (!c || !c->fused || c->world == 1) return direct_enqueue(c, buf, count, -1, static_cast<hipStream_t>(stream));
  XT_REQUIRE(buf, "xt_allreduce_direct: null argument");
  XT_REQUIRE(count > 0 && count <= c->max_count, "xt_allreduce_direct: count %lld outside (0, %lld]", (long long)count,
             (long long)c->max_count);
  XT_REQUIRE((reinterpret_cast<uintptr_t>(buf) & 15) == 0, "xt_allreduce_direct: the buffer must be 16-byte aligned");
  XT_REQUIRE(c->connected, "xt_allreduce_direct: xt_direct_connect has not been called");
  c->calls++;
  const int64_t nvec = (count + 3) / 4;
  // every block of every rank's launch must be resident at once (its blocks wait for OTHER ranks' blocks): the grid is capped
  // at (resident workgroups) / (ranks sharing this device) and the chunks are grid-strided
  unsigned g = (unsigned)((nvec + xt::kFusedVecs - 1) / xt::kFusedVecs);
  const unsigned cap = (unsigned)xt::spin_block_cap(c);
  if (g > cap) g = cap;
  hipLaunchKernelGGL(xt::xgmi_fused_kernel, dim3(g ? g : 1), dim3(256), 0, static_cast<hipStream_t>(stream), buf,
                     reinterpret_cast<const float*>(c->block + c->inbox_off),
                     reinterpret_cast<const float*>(c->block + c->result_off), c->slice_cap, count, c->rank, c->world, c->peers,
                     reinterpret_cast<const uint32_t*>(c->block), c->ctl, c->timeout_ticks);
  XT_LAUNCH_CHECK();
  return 0;
}

int xt_direct_set_fused(xt_direct_comm* c, int32_t fused) {
  XT_REQUIRE(c, "xt_direct_set_fused: null comm");
  c->fused = fused ? 1 : 0;
  return 0;
}

// N logical ranks driven by ONE host thread (in-process groups): the launches are issued phase by phase, so that no kernel
// ever waits for one that sits BEHIND it in a hardware queue (streams of one process share a handful of queues)
int xt_allreduce_direct_group(int32_t n, xt_direct_comm* const* comms, float* const* bufs, int64_t count, void* const* streams) {
  XT_REQUIRE(n >= 1 && comms && bufs && streams, "xt_allreduce_direct_group: null argument");
  for (int phase = 0; phase < 3; ++phase)
    for (int r = 0; r < n; ++r)
      if (int rc = direct_enqueue(comms[r], bufs[r], count, phase, static_cast<hipStream_t>(streams[r]))) return rc;
  return 0;
}

int xt_direct_exchange_hook(float* grads, int64_t count, void* user, void* stream) {
  return xt_allreduce_direct(static_cast<xt_direct_comm*>(user), grads, count, stream);
}

int xt_direct_set_timeout_ms(xt_direct_comm* c, int32_t ms) {
  XT_REQUIRE(c && ms > 0, "xt_direct_set_timeout_ms: bad argument");
  c->timeout_ticks = (unsigned long long)ms * 100000ull;
  return 0;
}

int xt_direct_status(xt_direct_comm* c, int32_t* calls, int32_t* seq, int32_t* error_bits) {
  XT_REQUIRE(c, "xt_direct_status: null comm");
  uint32_t w[2] = {0, 0};
  XT_CHECK_HIP(hipDeviceSynchronize());
  XT_CHECK_HIP(hipMemcpy(w, c->ctl, sizeof(w), hipMemcpyDeviceToHost));     // (synchronises with the null stream only)
  if (calls) *calls = c->calls;
  if (seq) *seq = (int32_t)w[0];
  if (error_bits) *error_bits = (int32_t)w[1];
  return 0;
}

int xt_direct_info(xt_direct_comm* c, int32_t* ranks_on_device, int32_t* block_cap) {
  XT_REQUIRE(c, "xt_direct_info: null comm");
  if (ranks_on_device) *ranks_on_device = c->ranks_on_device;
  if (block_cap) *block_cap = xt::spin_block_cap(c);
  return 0;
}

int xt_direct_read_result(xt_direct_comm* c, float* host_out, int64_t count) {
  XT_REQUIRE(c && host_out && count > 0 && count <= c->max_count, "xt_direct_read_result: bad argument");
  XT_CHECK_HIP(hipDeviceSynchronize());
  XT_CHECK_HIP(hipMemcpy(host_out, c->block + c->result_off, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost));
  return 0;
}

int xt_direct_reset(xt_direct_comm* c) {
  XT_REQUIRE(c, "xt_direct_reset: null comm");
  XT_CHECK_HIP(hipDeviceSynchronize());
  XT_CHECK_HIP(hipMemset(c->ctl, 0, sizeof(uint32_t) * xt::kCtlWords));
  XT_CHECK_HIP(hipMemset(c->block, 0, (size_t)xt::kIdentOff * 4));          // ready / done flags (the identity words stay)
  XT_CHECK_HIP(hipDeviceSynchronize());
  return 0;
}

int xt_direct_destroy(xt_direct_comm* c) {
  if (!c) return 0;
  (void)hipDeviceSynchronize();
  for (int q = 0; q < c->world; ++q)
    if (c->peer_ipc[q] && c->peer_block[q]) (void)hipIpcCloseMemHandle(c->peer_block[q]);
  if (c->block) (void)hipFree(c->block);
  if (c->ctl) (void)hipFree(c->ctl);
  (void)hipGetLastError();
  delete c;
  return 0;
}

}  // extern "C"

namespace xt {
// the scatter phase alone, for a step whose gradient did not come out of grads_finish_kernel (an EMPTY trajectory shard
// contributes zeros): copies buf -> the owners' inboxes and raises the ready flags; the step's reduce / optimiser launches
// follow as usual (the tickets are per phase and re-armed by the optimiser kernel)
int direct_launch_scatter(xt_direct_comm* c, const float* buf, int64_t count, hipStream_t st) {
  XT_REQUIRE(c && c->connected && buf && count > 0 && count <= c->max_count, "direct exchange (scatter): bad comm / count");
  const int64_t nvec = (count + 3) / 4;
  const int64_t slice_max = (nvec + c->world - 1) / c->world;
  const unsigned sb = (unsigned)((slice_max + kScatterVecs - 1) / kScatterVecs);
  hipLaunchKernelGGL(xgmi_scatter_kernel, dim3(sb ? sb : 1, c->world), dim3(256), 0, st, buf, count, c->rank, c->world,
                     c->peers, c->ctl);
  XT_LAUNCH_CHECK();
  return 0;
}
}  // namespace xt
