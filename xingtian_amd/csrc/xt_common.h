// Shared host/device helpers for the MI355X (gfx950) learner kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "xt_mi355x.h"

namespace xt {

void set_error(const char* fmt, ...);
xt_tuning& tuning();
int& last_arith();             // XT_ARITH_* of the most recent layer launch (diagnostic)      // process-wide kernel-selection knobs (xt_tuning_get / xt_tuning_set); defined in xt_net.hip

#define XT_CHECK_HIP(expr)                                                                  \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      (void)hipGetLastError(); /* do not leave a sticky error for the host framework */      \
      xt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return 1;                                                                             \
    }                                                                                       \
  } while (0)

#define XT_REQUIRE(cond, ...)          \
  do {                                 \
    if (!(cond)) {                     \
      xt::set_error(__VA_ARGS__);      \
      return 2;                        \
    }                                  \
  } while (0)

#define XT_LAUNCH_CHECK() XT_CHECK_HIP(hipGetLastError())

// ---- per-block timeline instrumentation (diagnostic builds only: make TL=1 -> libxt_mi355x_tl.so).
// Every block's thread 0 stores the 100 MHz wall clock at up to 6 marks + a role word + HW_ID/XCC_ID into a
// host-provided buffer [block][8] (tools/timeline.py); compiled out of the product library.
#ifdef XT_TIMELINE
static __device__ unsigned long long* xt_tl_ptr;
#define XT_TL_BLOCK() ((size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z))
#define XT_TL(slot)                                                                                   \
  do {                                                                                                \
    if (threadIdx.x == 0 && xt_tl_ptr) xt_tl_ptr[XT_TL_BLOCK() * 8 + (slot)] = wall_clock64();        \
  } while (0)
#define XT_TL_DRAIN(slot)                                     \
  do {                                                        \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          \
    XT_TL(slot);                                              \
  } while (0)
#define XT_TL_ROLE(role)                                                                              \
  do {                                                                                                \
    if (threadIdx.x == 0 && xt_tl_ptr) {                                                              \
      unsigned hw, xcc;                                                                               \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                              \
      xt_tl_ptr[XT_TL_BLOCK() * 8 + 6] = (unsigned long long)(role);                                  \
      xt_tl_ptr[XT_TL_BLOCK() * 8 + 7] = ((unsigned long long)xcc << 32) | hw;                        \
    }                                                                                                 \
  } while (0)
#define XT_TL_SETTER(name)                                                                            \
  extern "C" int xt_tl_set_##name(void* p) {                                                          \
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(xt_tl_ptr), &p, sizeof(p));                              \
  }
#else
#define XT_TL(slot) do {} while (0)
#define XT_TL_DRAIN(slot) do {} while (0)
#define XT_TL_ROLE(role) do {} while (0)
#define XT_TL_SETTER(name)
#endif

// WRITE-THROUGH stores (sc1): a large output that the next kernel reads goes to memory while the rest of the launch
// still runs, instead of sitting dirty in the L2 until the end-of-kernel release writes it back -- the dependent kernel
// boundary costs + (dirty bytes / ~6 TB/s) otherwise (MI355X_MICROARCH.md, rows "boundary" / "publish-large").
// Round 4: both forms are COMPILER-VISIBLE, no inline asm.  The 16-byte store is a raw buffer store with aux = sc1
// through a descriptor of the output tensor (cdna_hip_programming.md, the publish recipe); the 4-byte one a relaxed
// agent-scope atomic store, which gfx950 lowers to `global_store_dword ... sc1`.  Round 3 used
// `asm volatile("global_store_dwordx4 ... sc1\n\ts_nop 1")`: on gfx940+ a VMEM store of more than 64 bits must be followed
// by two wait states before a VALU instruction overwrites its data registers (LLVM GCNHazardRecognizer::
// checkVALUHazardsHelper, VALUWaitStates = 2 with GFX940 instructions), and the hazard recogniser does not look inside an
// asm statement -- correctness hung on a hand-placed s_nop.  Now the recogniser sees the store.
// (__builtin_nontemporal_store sets nt instead, which measured WORSE than plain stores: 6.92 vs 6.89 ms per update
// against 6.77 for sc1.)  -DXT_NO_WT (the `nowt` twin library, tests/test_gpu_wt_stress.py) compiles both to plain stores.
typedef uint32_t xt_u32x4 __attribute__((ext_vector_type(4)));
constexpr int kAuxSc1 = 16;         // gfx940+ cache-policy bits of buffer instructions: 1 = sc0, 2 = nt, 16 = sc1
__device__ __forceinline__ void store1_wt(float* base, size_t off, const float v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(XT_NO_WT)
  __hip_atomic_store(base + off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  base[off] = v;
#endif
}
// `base`: wave-uniform tensor base (it becomes the descriptor in SGPRs), `off`: element offset < 2^29 (the ABI bounds
// every activation / gradient tensor by 2 GiB, include/xt_mi355x.h: xt_layer_dgrad)
__device__ __forceinline__ void store4_wt(float* base, size_t off, const float4 v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(XT_NO_WT)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  xt_u32x4 q;
  q.x = __float_as_uint(v.x); q.y = __float_as_uint(v.y); q.z = __float_as_uint(v.z); q.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(q, rs, (int)(off * 4), 0, kAuxSc1);
#else
  *reinterpret_cast<float4*>(base + off) = v;
#endif
}

// n / d for n*d < 2^32 via one v_mul_hi_u32 (magic = floor(2^32/d)+1).
struct FastDiv {
  uint32_t d, magic;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  f.magic = (d <= 1) ? 0u : (uint32_t)((1ull << 32) / d) + 1u;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) {
  return f.d <= 1 ? n : __umulhi(n, f.magic);
}

// The transcendental activations are kept OUT of line (one shared body) so that epilogues that apply the activation to
// 16..64 accumulator registers do not inline ~60 instructions of libdevice code 64 times (code size = cold
// instruction-fetch time per launch).  ACTIVATION_MAP of xt/model/model_utils.py:8-20; swish and gelu are not
// monotonic: for them act_grad is given the PRE-activation (see act_needs_preact).
constexpr float kSeluScale = 1.0507009873554805f, kSeluAlpha = 1.6732632423543772f;   // tf.nn.selu
constexpr float kLeakyAlpha = 0.2f;                                                  // tf.nn.leaky_relu default
__device__ __noinline__ static float xt_act_slow(float z, int act) {
  switch (act) {
    case XT_ACT_TANH: return tanhf(z);
    case XT_ACT_SIGMOID: return 1.f / (1.f + expf(-z));
    case XT_ACT_SOFTSIGN: return z / (1.f + fabsf(z));
    case XT_ACT_SOFTPLUS: return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z)));       // log(1 + e^z), overflow-free
    case XT_ACT_ELU: return z > 0.f ? z : expm1f(z);
    case XT_ACT_SELU: return kSeluScale * (z > 0.f ? z : kSeluAlpha * expm1f(z));
    case XT_ACT_SWISH: return z / (1.f + expf(-z));
    case XT_ACT_GELU: return 0.5f * z * (1.f + tanhf(0.7978845608028654f * (z + 0.044715f * z * z * z)));
    default: return z;
  }
}
__device__ __noinline__ static float xt_act_grad_slow(float y, int act) {
  switch (act) {
    case XT_ACT_TANH: return 1.f - y * y;
    case XT_ACT_SIGMOID: return y * (1.f - y);
    case XT_ACT_SOFTSIGN: { const float u = 1.f - fabsf(y); return u * u; }     // 1 - |y| = 1 / (1 + |z|)
    case XT_ACT_SOFTPLUS: return -expm1f(-y);                                   // sigmoid(z) = 1 - e^{-y}
    case XT_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;                            // TF EluGrad (from the outputs)
    case XT_ACT_SELU: return y > 0.f ? kSeluScale : y + kSeluScale * kSeluAlpha; // TF SeluGrad (from the outputs)
    // not monotonic: `y` IS the pre-activation z for these two (act_needs_preact)
    case XT_ACT_SWISH: { const float sg = 1.f / (1.f + expf(-y)); return sg + y * sg * (1.f - sg); }
    case XT_ACT_GELU: {
      const float c = 0.7978845608028654f, th = tanhf(c * (y + 0.044715f * y * y * y));
      return 0.5f * (1.f + th) + 0.5f * y * (1.f - th * th) * c * (1.f + 3.f * 0.044715f * y * y);
    }
    default: return 1.f;
  }
}

// activations whose derivative cannot be taken from the output: the producer's PRE-activation is kept and handed to
// act_grad instead (xt_net: Layer::z_off)
static inline bool act_needs_preact(int act) { return act == XT_ACT_SWISH || act == XT_ACT_GELU; }

__device__ __forceinline__ float act_apply(float z, int act) {
  if (act == XT_ACT_RELU) return z > 0.f ? z : 0.f;
  if (act == XT_ACT_NONE) return z;
  if (act == XT_ACT_LEAKY_RELU) return z > 0.f ? z : kLeakyAlpha * z;
  return xt_act_slow(z, act);
}
// d(post)/d(pre-activation) from the saved OUTPUT y (every supported activation is monotonic)
__device__ __forceinline__ float act_grad(float y, int act) {
  if (act == XT_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == XT_ACT_NONE) return 1.f;
  if (act == XT_ACT_TANH) return 1.f - y * y;
  if (act == XT_ACT_LEAKY_RELU) return y > 0.f ? 1.f : kLeakyAlpha;
  return xt_act_grad_slow(y, act);
}

// ---- argument blocks shared between translation units -------------------------------------
struct LossArgs {
  const float* terms;   // [B,4] per-sample surr, entropy, vf ; nullptr -> skip
  int B;
  float ent_coef, critic_coef, inv_b;
  float* out;           // 4 floats (may be null)
  float* acc;           // running sum / count (may be null)
  // IMPALA's sum-form loss (impala_cnn_opt.py:351): per-trajectory sums written by the v-trace kernel, added up
  // in trajectory order; used when terms == nullptr and traj_loss != nullptr
  const float* traj_loss;
  int n_traj;
  int acc_set;          // != 0: this is the FIRST chunk of a train -- acc = {loss, 1} instead of +=: no memset node in front of it
};

// One entry per parameter block: sum `nslab` partial slabs (fixed order) into dst and accumulate the
// squared norm of the result.  nslab == 1 with src == dst only accumulates the norm.
struct GradEntry {
  const float* src;
  float* dst;
  int count, nslab;
  long long stride;
  int zl;               // z lanes per block (power of two <= 32)
  int blk0, nblk;       // first block of the entry in THIS launch, number of blocks
  int pre;              // > 0: the producing kernel has written the entry's `pre` squared-norm partials already (and dst is
                        // final): the entry only reserves its slots (and, in the fused tail, its blocks only apply the update)
  int pblk0;            // slot of the entry's first squared-norm partial (position in the full table's block order: the same
                        // whether the table is reduced by one launch or by two partial ones)
};
struct GradTable {
  int n;
  GradEntry e[12];
};

struct PpoHeadArgs {
  const float *f_pi, *f_v, *wpi, *bpi, *wv, *bv;
  const int32_t *idx, *action;
  const float *old_logp, *old_v;
  const double *adv, *target_v;
  float clip_ratio, ent_coef, vf_clip, critic_coef, inv_b;
  int B, F, A, act_prev, shared;
  float *logits, *value, *dlogits, *dvalue, *terms, *df_pi, *df_v;
  // deferred split-K finish of the last trunk layer(s): feature = act(sum_z part[z] + bias); written to
  // feat_*_w (the layer's activation buffer) by this kernel.  part_* == nullptr -> features are final already.
  const float *part_pi, *part_v, *tbias_pi, *tbias_v;
  float *feat_pi_w, *feat_v_w;
  int ksplit_pi, ksplit_v, act_feat;
  long long part_stride;
};

// IMPALA (ImpalaCnnOpt, one shared trunk): heads forward with the deferred split-K finish of the last trunk layer
struct ImpalaHeadArgs {
  const float *feat, *wpi, *bpi, *wv, *bv;
  const float *part, *tbias;   // part != nullptr: feature = act(sum_z part[z] + tbias), written to feat_w
  float* feat_w;
  int ksplit, act_feat;
  long long part_stride;
  int B, F, A;
  float *logits, *value;
};
// v-trace targets + sum-form loss + d(logits, baseline) + d(features) of one chunk, one workgroup per trajectory
struct ImpalaLossArgs {
  const float *logits, *baseline, *bp_logits;
  const int32_t* action;
  const uint8_t* done;
  const float* reward;
  int T, A, F, act_prev;
  float gamma;
  float *dlogits, *dbaseline, *traj_loss, *vs_out, *pg_out;
  const float *feat, *wpi, *wv;
  float* dfeat;
};

// norm finalisation executed by the last block of grads_finish_kernel (ticket counter)
struct FinalizeArgs {
  int enable;                // 1: last block finalises (ticket); 2: an extra block does the norm-independent scalars;
                             // 3: as 2, and the launch ALSO applies Adam behind a grid barrier (fused tail, `ap`)
  unsigned int* counter;     // zero before the first launch; the last block resets it
  float clip_norm, grad_scale, lr, beta1, beta2;
  float* state;
  LossArgs loss;
  const float* lr_dev;       // != nullptr: the step size is read from device memory (lr_schedule inside a replayed hipGraph)
  struct {                   // enable == 3
    float *params, *m, *v;
    const float* grads;      // base of the flat gradient buffer (entry dst - grads = offset into params / m / v)
    float eps;
    int npartials;           // squared-norm partial slots of the whole table
  } ap;
};

// ---- data-parallel step (xingtian_amd/parallel.py::LearnerDP; the reference's only analogue is the dead host-side
// trainer, xt/framework/trainer.py:86-92,139-144).  The exchanged buffer is the flat gradient followed by a TAIL of
// XT_DP_TAIL_FLOATS = 2 x 16 slots: [r] = rows rank r held for this update, [16 + r] = rank r's share of the step's loss.
// Every rank fills its own two slots and zeroes the others, so after the SUM all-reduce every rank holds every rank's
// values EXACTLY (one non-zero summand per slot) and derives the same global loss in rank order -- no host collective.
constexpr int kDpMaxWorld = 16;
constexpr int kDpTailFloats = XT_DP_TAIL_FLOATS;
constexpr int kDpRedBlocksMax = 128;          // workgroups of the direct exchange's reduce launch (per rank)

struct DirectPeers {
  float* inbox_me[kDpMaxWorld];     // peer q's inbox slot for THIS rank
  float* result[kDpMaxWorld];       // peer q's result buffer
  uint32_t* flags[kDpMaxWorld];     // peer q's flag words
  float* norm_part[kDpMaxWorld];    // peer q's squared-norm partials [world][reduce blocks]
};

struct DpFinish {                   // what grads_finish_kernel does for a data-parallel step
  float* tail;                      // != nullptr: the extra block writes this rank's tail slots here (plain exchange)
  int rank, world;
  float rows;
  // direct exchange fused into the step: every reduced float4 goes straight into the owning peer's inbox (no gradient
  // buffer round trip, no scatter launch); the last block of the launch raises the ready flags (FinalizeArgs.counter)
  int scatter;
  const float* grads_base;          // flat index of an entry = dst - grads_base
  long long nvec;                   // float4s of the exchanged buffer (gradient + tail)
  DirectPeers peers;
  uint32_t* ctl;                    // the comm's device-local control words
};

struct DpStep {                     // what the optimiser kernels do for a data-parallel step
  const float* tail;                // the EXCHANGED tail (nullptr: none): block 0 adds the global loss to acc, checks the rows
  int world;
  float loss_scale;                 // PPO weak mode: 1 / world (mean of the ranks' means); else 1 (shares of one sum)
  float* acc;                       // [0] += loss, [1] += 1, [2] = error bits
  // direct exchange fused into the step: ONE small launch in front of the optimiser (dp_reduce_wait_kernel, red_blocks <= 128
  // workgroups) waits for every rank's scatter, sums this rank's slice over the inbox slots in fixed rank order, pushes
  // the slice and its squared-norm partials to every peer, raises the done flags and waits for everybody's; the optimiser
  // launch then reads gradient and partials out of the exchange block without spinning, its last block re-arms the comm
  const uint32_t* flags;            // my flag words (nullptr: plain exchange, nothing to wait for)
  uint32_t* ctl;
  unsigned long long timeout_ticks;
  const float* inbox;               // my inbox: world slots of slice_cap floats
  long long slice_cap, nvec, nvec_grad;
  int rank, red_blocks;
  DirectPeers peers;
};

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// One-time set-up PER DEVICE (function attributes such as the dynamic-LDS limit belong to the device's copy of the
// code object): a process may drive several GPUs, one learner thread each (PBT-style multi-learner set-ups).  The body
// must be idempotent: two threads on the same device may both run it.
struct PerDeviceOnce {
  std::atomic<uint64_t> done{0};
  template <typename F>
  void run(F body) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    body();
    done.fetch_or(bit, std::memory_order_release);
  }
};

// xt_train_io.tail_in_graph: what the learner thread and the tail kernels of a train say to each other (page-locked host memory)
struct IoMailbox {
  unsigned long long publish_dst;   // host -> device: device-side address the new parameters go to (0: nobody asked)
  uint32_t seq;                     // host -> device: sequence number of the train being launched
  uint32_t pad0;
  float loss[4];                    // device -> host: loss_acc of that train
  uint32_t loss_seq;                // device -> host: == seq once loss[] has landed
  uint32_t publish_seq;             // device -> host: sequence number of the last train whose parameter copy has landed
  uint32_t snap_seq;                // device -> host: ... whose parameter SNAPSHOT is complete in device memory (mode 2)
  uint32_t pad1[5];
};
static_assert(sizeof(IoMailbox) == 64, "IoMailbox is one 64-byte line");
// ... and the same tail FOLDED into the Adam kernel of the train's last chunk (tail_in_graph = 2, Adam, no gradient exchange): its
// block 0 reports the loss before it updates anything, every block writes its updated parameters to the snapshot buffer as
// well (system-scope write-through), the last block to finish reports the snapshot -- no kernel behind the optimiser at all
struct IoFold {
  IoMailbox* mb;               // device-side address of the mailbox
  float* acc;                  // the library-owned loss accumulator (re-armed here)
  float* loss_out;             // the caller's device-side loss_acc
  unsigned long long* fwd;     // [0] destination announced, [1] sequence number, [2] block ticket
  float* snap;                 // the snapshot buffer of this train's parity
};
// xt_sdma.hip: a device -> page-locked-host copy on the SDMA engine through the process's HSA runtime, synchronous; -> nullptr
// or why this process cannot do it.  `sig_io`: an hsa_signal_t handle kept by the caller (0 = create one)
const char* sdma_copy_d2h(void* dst_host, const void* src_dev, size_t bytes, unsigned long long* sig_io);
void sdma_signal_destroy(unsigned long long* sig);
}  // namespace xt
