// Network-level driver: one C call enqueues a whole Model.train() (all epochs and
// minibatches of xt/model/ppo/ppo.py:111-132, or one ImpalaCnnOpt.train chunk) on a HIP
// stream, optionally captured once into a hipGraph and replayed (a B=320 step is ~20
// short kernels; the reference pays a feed_dict H2D + session dispatch per minibatch).
#include <atomic>
#include <chrono>
#include <thread>
#include <mutex>
#include <vector>
#include <string>
#include <string.h>
#include <stdlib.h>

#include "xt_common.h"
#include "xt_heads_dev.h"

namespace xt {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int& last_arith() { static thread_local int a = 0; return a; }      // per calling thread (one learner thread per GPU)
xt_tuning& tuning() {
  static xt_tuning t = {/*bf16x6*/ 1, /*dgrad_all_classes*/ 1, /*dgrad_tile64*/ 1, /*dgrad_halo*/ 1, /*bwd_own_instance*/ 1,
                        /*bwd_fit_slots*/ 768, /*conv1_bf16x3*/ 1, /*conv1_flat*/ 1, /*conv1_waves*/ 8,
                        /*fwd_two_groups*/ 1, /*direct*/ 1, /*direct_fwd*/ 1, /*direct_dgrad*/ 1, /*direct_all*/ 0,
                        /*direct_waves*/ 1536, /*direct_max_waves*/ 8, /*direct_tile64_tiles*/ 3072,
                        /*fwd_split_target*/ 256, /*wgrad_split_target*/ 512, /*reduce_z_lanes*/ 8,
                        /*defer_splitk*/ 1, /*finalize_ticket*/ 0, /*fwd_tiled_valid*/ 1, /*wgrad_rows*/ 4, /*fwd_prefetch_all*/ 0, /*bwd_deep_prefetch*/ 1, /*fwd_four_groups*/ 1, /*reduce_deep_lanes*/ 128, /*fwd_xcd_chunk*/ 1,
                        /*tail_overlap*/ 0, /*tail_fused*/ 0, /*dense_wgrad_x6*/ 1, /*fwd_fuse12*/ 0};
  return t;
}

int launch_fwd(const xt_conv_geom*, const xt_input_xform*, int, const void*, const int32_t*, const float*,
               const float*, float*, float*, int, hipStream_t, int* deferred_ksplit = nullptr,
               uint32_t* relu_mask = nullptr, int* mask_written = nullptr);
int launch_bwd_layer(const xt_conv_geom*, int, const float*, const float*, const float*, int, float*, float*, float*,
                     int, const HeadWgArgs*, int*, hipStream_t, const uint32_t* xmask = nullptr, int slab_cap = 0,
                     const float* x_grad = nullptr, float* sq_partials = nullptr, int* npre_out = nullptr);
int launch_act_apply(const float* z, float* y, long long count, int act, hipStream_t st);
int launch_conv12_same_fwd(const xt_conv_geom*, const xt_input_xform*, const xt_conv_geom*, int, const void*, const int32_t*,
                           const float*, const float*, float*, const float*, const float*, float*, hipStream_t);
int launch_wgrad(const xt_conv_geom*, const xt_input_xform*, int, const void*, const int32_t*, const float*,
                 float*, float*, int, hipStream_t, int reduce_now = 1, int* msplit_out = nullptr, int slab_cap = 0);
int launch_dgrad(const xt_conv_geom*, int, const float*, const float*, const float*, int, float*, hipStream_t);
int launch_global_norm(const float*, long long, float, float, float, float, float, int, float*, float*, hipStream_t,
                       const float* lr_dev = nullptr, int* nblocks_out = nullptr);
int launch_grads_finish(GradTable*, float*, int, int*, const FinalizeArgs*, hipStream_t, unsigned select = 0,
                        unsigned early = 0, const DpFinish* dpf = nullptr);
int launch_dp_tail_write(float*, int, float, const float*, float*, float, const float*, float, float, int, hipStream_t);
int launch_dp_tail_consume(const DpStep*, hipStream_t);
int launch_dp_reduce_wait(const DpStep*, hipStream_t);
// the direct exchange fused into the step (xt_xgmi.hip)
int direct_fill_finish(xt_direct_comm*, int64_t, DpFinish*);
int direct_launch_scatter(xt_direct_comm*, const float*, int64_t, hipStream_t);
int direct_fill_step(xt_direct_comm*, int64_t, int64_t, DpStep*, const float**, const float**, int*, int*);
int grads_finish_resident_blocks();
int grads_finish_fused_grid(const GradTable*);
int launch_sqnorm_partial(const float*, long long, float*, int*, hipStream_t);
int launch_norm_finalize(const float*, int, float, float, float, float, float, int, float*, const LossArgs*, hipStream_t,
                         const float* lr_dev = nullptr);
int launch_ppo_heads_fused(const PpoHeadArgs&, hipStream_t);
int launch_adam_clip(float*, const float*, float*, float*, long long, float, float, float, float*, const float*, int,
                     float, float, hipStream_t, const DpStep* dp = nullptr, int block_cap = 0, const IoFold* io = nullptr);
int launch_rmsprop_clip(float*, const float*, float*, float*, long long, float, float, float, float*, const float*, int,
                        float, float, hipStream_t, const float* lr_dev = nullptr, const DpStep* dp = nullptr,
                        int block_cap = 0);
int launch_impala_heads_fwd(const ImpalaHeadArgs&, hipStream_t);
int launch_impala_vtrace_bwd(const ImpalaLossArgs&, int, hipStream_t);
int launch_impala_loss_reduce(const float*, int, float*, float*, hipStream_t);
int launch_ppo_loss_gauss(const float*, const float*, const float*, int, int, const int32_t*, const float*, const float*,
                          const double*, const float*, const double*, float, float, float, float, float, float*, float*,
                          float*, int, float*, hipStream_t);
int launch_heads_dfeat(const float*, const float*, int, int, int, const float*, const float*, const float*,
                       const float*, int, float*, float*, hipStream_t);
int launch_heads_wgrad_partial(const float*, const float*, int, int, int, const float*, const float*, float*,
                               long long, float*, long long, int*, hipStream_t);
int launch_adam(float*, const float*, float*, float*, long long, float, float, float, const float*, hipStream_t);

struct Layer {
  xt_conv_geom g;
  int64_t poff;
  int trunk;
  int K, OHOW;
  int64_t act_off, dact_off;   // floats into the workspace
  int64_t slab_off;            // this layer's wgrad slabs
  int64_t part_off;            // split-K partials of a trunk's last layer (deferred finish), else -1
  int last_msplit, last_ksplit;
  int last_npre = 0;      // squared-norm partials the last backward launch of this layer left in the norm scratch (0: none)
  int slab_cap;                // number of slabs the region can hold
  int64_t mask_off;            // relu sign mask of this layer's output (one word per position), else -1
  int mask_valid;              // the last forward of this layer wrote the mask
  int64_t z_off;               // pre-activation of this layer (swish / gelu: act_needs_preact), else -1
};

static inline int64_t align4(int64_t x) { return (x + 3) & ~int64_t(3); }

}  // namespace xt

struct xt_net {
  std::vector<xt::Layer> layers;
  int t_begin[2], t_end[2];
  int n_trunks, feat, A;
  int64_t pi_off, v_off, P;
  xt_input_xform xf;
  int in_h, in_w, in_c;
  int maxB;
  // bound buffers
  float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr, *state = nullptr;
  float* ws = nullptr;
  int64_t ws_floats = 0;
  // workspace carve (float offsets)
  int64_t off_counter;
  int64_t off_partial, off_logits, off_value, off_dlogits, off_dvalue, off_terms, off_loss, off_norm;
  int action_type = 0;          // XT_ACTION_*
  int64_t logstd_off = 0;       // pi_logstd [A] in the flat parameter buffer (DiagGaussian)
  int64_t off_dls = 0;          // [maxB][align4(A)] per-sample d loss / d pi_logstd rows
  int dls_rows = 0;
  int64_t off_hslab_pi, off_hslab_v, hstride_pi, hstride_v;
  int64_t partial_floats;
  int head_chunks = 0, norm_blocks = 0;
  // graph cache for ppo_train
  xt_grad_exchange_fn xchg = nullptr;  // xt_net_set_grad_exchange
  void* xchg_user = nullptr;
  int xchg_flags = 0;                  // XT_XCHG_OVERLAP: two buckets, the first exchanged under the rest of the backward
  // xt_net_set_rccl: the exchange served by the library itself (a direct call of ncclAllReduce through the function
  // pointer the caller resolved, no host-language trampoline on the enqueue path)
  void* rccl_comm = nullptr;
  xt_nccl_allreduce_fn rccl_fn = nullptr;
  int rccl_calls = 0, rccl_last_error = 0;
  // xt_net_set_dp: the exchanged buffer carries a tail of rows / loss shares behind the gradient (no host collectives);
  // xt_net_set_direct: the direct exchange fused into the step (scatter inside the gradient reduction, gather inside the
  // optimiser kernel)
  int dp_rank = 0, dp_world = 0;
  float dp_loss_scale = 1.f, dp_rows = 0.f;
  xt_direct_comm* direct = nullptr;
  hipStream_t xchg_stream = nullptr;   // side stream of the first bucket's exchange
  hipEvent_t xchg_fork = nullptr, xchg_join = nullptr;
  // single-GPU tail overlap (xt_tuning.tail_overlap): a side stream for the first gradient bucket's slab reduction and
  // for the bulk of the optimiser update; adam_pending = the side stream's update has not been joined yet
  hipStream_t tail_stream = nullptr;
  hipEvent_t tail_fork = nullptr, tail_join = nullptr, adam_fork = nullptr, adam_join = nullptr;
  bool adam_pending = false;
  // hipGraph cache of the whole-update entry points: a few slots, because the streaming ingest alternates between
  // two rollout buffer sets (two pointer sets -> two graphs), least recently used replaced
  struct GraphSlot { std::string key; hipGraphExec_t exec = nullptr; unsigned long long used = 0; };
  GraphSlot gslots[8];
  unsigned long long gclock = 0;
  hipStream_t cap_stream = nullptr;   // capture happens on our own stream: the legacy null stream cannot be captured
  // xt_train_io.tail_in_graph: the 64-byte page-locked mailbox between the learner thread and the train's two tail kernels
  // (host side / the same block through the device's address space), the sequence number of the last train that used it, and
  // whether the train being enqueued carries the tail (part of the graph key)
  xt::IoMailbox* io_mb = nullptr;
  xt::IoMailbox* io_mb_dev = nullptr;
  uint32_t io_seq = 0;
  // tail_in_graph = 2: the 80 us bus-bound copy of the new parameters runs UNDER THE NEXT TRAIN -- the train's graph ends with
  // a device-side snapshot (params -> io_snap[seq & 1], ~5 us, reported through the mailbox), the copy snapshot -> page-locked
  // destination runs on the SDMA engine, issued through the HSA runtime by whoever waits for the publish (the weights ring's
  // committer thread: xt_net_io_publish_wait); no HIP stream or event is involved.  (A copy KERNEL beside the train does not
  // work, and this process's hipMemcpyAsync is one: xt_sdma.hip.)
  float* io_snap[2] = {nullptr, nullptr};
  // per snapshot buffer: the page-locked destination and the sequence number of the publish that owns it (0: none yet; written
  // by the learner thread before the launch), the publish some thread has CLAIMED the copy of, the publish whose copy has landed
  void* io_dst[2] = {nullptr, nullptr};
  std::atomic<uint32_t> io_snap_owner[2] = {{0}, {0}}, io_copying[2] = {{0}, {0}}, io_copied[2] = {{0}, {0}};
  unsigned long long io_sig[2] = {0, 0};          // hsa_signal_t handles of the copies (xt_sdma.hip), created on first use
  const xt::IoFold* io_fold = nullptr;  // set while the LAST chunk of a tail_in_graph = 2 train is enqueued: its Adam kernel carries the tail
  bool io_acc_clean = false;          // the library-owned loss accumulator of tail_in_graph trains is zero (its loss kernel re-arms it)
  double io_us[4] = {0, 0, 0, 0};     // xt_net_io_times: host time of xt_net_impala_train_io's phases, accumulated
  long long io_calls = 0;
  int64_t off_iofwd = 0;              // 4 floats of the workspace: {destination (64 bit), sequence number} handed kernel to kernel
  int64_t off_ioacc = 0;              // 4 floats: the loss accumulator of tail_in_graph trains (no memset node in their graph)
};

namespace xt {

static int fwd_split(const Layer& L, int B) {
  const int M = B * L.OHOW, N = L.g.N;
  const int tiles = (N <= 32) ? ((M + 127) / 128) * ((N + 31) / 32) : ((M + 63) / 64) * ((N + 63) / 64);
  const int ksteps = (L.K + 31) / 32;
  if (tiles >= 128 || ksteps < 4) return 1;
  // note: the fused PPO head kernel finishes at most 16 partial slabs (kMaxHeadSplit)
  const int target = tuning().fwd_split_target;    // measured: 256 beats 512/128 (Dense 3136->256: 16.6 vs 20.7 us)
  int s = target / tiles;
  if (s > ksteps / 2) s = ksteps / 2;
  if (s > 16) s = 16;
  return s < 1 ? 1 : s;
}
static int wgrad_split(const Layer& L, int B) {
  const int M = B * L.OHOW, N = L.g.N;
  const int tiles = (N <= 32) ? ((L.K + 127) / 128) * ((N + 31) / 32) : ((L.K + 63) / 64) * ((N + 63) / 64);
  const int msteps = (M + 31) / 32;
  const int wtarget = tuning().wgrad_split_target;
  int s = wtarget / tiles;
  if (s > msteps / 4) s = msteps / 4;
  return s < 1 ? 1 : s;
}

// xt_tuning.tail_overlap applies to one-trunk nets with >= 2 layers whose first layer opens the flat parameter buffer
static int tail_overlap_mode(xt_net* n) {
  const int t = tuning().tail_overlap;
  if (!t || n->n_trunks != 1 || n->layers.size() < 2 || n->layers[0].poff != 0 || tuning().finalize_ticket) return 0;
  const int64_t off_a = n->layers.back().poff;
  if (!(off_a > 0 && n->pi_off > off_a && n->v_off > off_a)) return 0;
  if (!n->tail_stream) {
    if (hipStreamCreateWithFlags(&n->tail_stream, hipStreamNonBlocking) != hipSuccess) { n->tail_stream = nullptr; return 0; }
    hipEvent_t* ev[4] = {&n->tail_fork, &n->tail_join, &n->adam_fork, &n->adam_join};
    for (hipEvent_t* e : ev)
      if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return 0;
  }
  return t & 3;
}

// the side stream's share of the previous optimiser update must have landed before anything but the first layer
// reads the parameters
static int join_pending_update(xt_net* n, hipStream_t st) {
  if (!n->adam_pending) return 0;
  XT_CHECK_HIP(hipStreamWaitEvent(st, n->adam_join, 0));
  n->adam_pending = false;
  return 0;
}

// defer_last: leave the split-K partials of every trunk's LAST layer un-finished (the fused PPO head kernel
// sums them); each such layer has its own partial region.
static int net_forward(xt_net* n, const void* obs, const int32_t* idx, int B, bool with_heads, hipStream_t st,
                       bool defer_last = false) {
  for (int tr = 0; tr < n->n_trunks; ++tr) {
    const void* x = obs;
    const int l0 = n->t_begin[tr];
    for (int l = l0; l < n->t_end[tr]; ++l) {
      Layer& L = n->layers[l];
      const bool first = (l == n->t_begin[tr]);
      const bool defer = defer_last && (l == n->t_end[tr] - 1) && L.part_off >= 0 && L.z_off < 0;
      L.last_ksplit = 1;
      L.mask_valid = 0;
      if (!first) { if (int rc = join_pending_update(n, st)) return rc; }
      if (first && tuning().fwd_fuse12 > 0 && B <= tuning().fwd_fuse12 && l + 2 < n->t_end[tr] && L.z_off < 0 &&
          L.mask_off < 0 && n->layers[l + 1].z_off < 0 && n->layers[l + 1].mask_off < 0) {
        // ImpalaCnnOpt 84x84 at a few hundred frames: conv1 -> conv2 of a frame stack in ONE launch (conv1's output stays in
        // LDS for conv2; both activations still go to their workspace buffers for the backward pass).  -1: not that geometry
        Layer& L2 = n->layers[l + 1];
        if (int rcj = join_pending_update(n, st)) return rcj;      // (the launch reads the second layer's weights too)
        const int rc = launch_conv12_same_fwd(&L.g, &n->xf, &L2.g, B, x, idx, n->params + L.poff,
                                              n->params + L.poff + (int64_t)L.K * L.g.N, n->ws + L.act_off, n->params + L2.poff,
                                              n->params + L2.poff + (int64_t)L2.K * L2.g.N, n->ws + L2.act_off, st);
        if (rc > 0) return rc;
        if (rc == 0) {
          L2.last_ksplit = 1;
          L2.mask_valid = 0;
          x = n->ws + L2.act_off;
          ++l;
          continue;
        }
      }
      if (L.z_off >= 0) {
        // swish / gelu: the layer writes its PRE-activation (the backward pass needs it), then one elementwise launch
        // produces the output the next layer reads -- a slow path, taken by no bundled configuration
        xt_conv_geom gz = L.g;
        gz.act = XT_ACT_NONE;
        if (int rc = launch_fwd(&gz, first ? &n->xf : nullptr, B, x, first ? idx : nullptr, n->params + L.poff,
                                n->params + L.poff + (int64_t)L.K * L.g.N, n->ws + L.z_off, n->ws + n->off_partial,
                                fwd_split(L, B), st, nullptr, nullptr, nullptr))
          return rc;
        if (int rc = launch_act_apply(n->ws + L.z_off, n->ws + L.act_off, (long long)B * L.OHOW * L.g.N, L.g.act, st)) return rc;
        x = n->ws + L.act_off;
        continue;
      }
      if (int rc = launch_fwd(&L.g, first ? &n->xf : nullptr, B, x, first ? idx : nullptr, n->params + L.poff,
                              n->params + L.poff + (int64_t)L.K * L.g.N, n->ws + L.act_off,
                              n->ws + (defer ? L.part_off : n->off_partial), fwd_split(L, B), st,
                              defer ? &L.last_ksplit : nullptr,
                              L.mask_off >= 0 ? reinterpret_cast<uint32_t*>(n->ws + L.mask_off) : nullptr, &L.mask_valid))
        return rc;
      x = n->ws + L.act_off;
    }
  }
  if (int rc = join_pending_update(n, st)) return rc;     // (one-layer trunks: before the heads read their weights)
  if (!with_heads) return 0;
  const float* f_pi = n->ws + n->layers[n->t_end[0] - 1].act_off;
  const float* f_v = n->ws + n->layers[n->t_end[n->n_trunks - 1] - 1].act_off;
  const int F = n->feat, A = n->A;
  return xt_heads_fwd(f_pi, f_v, B, F, A, n->params + n->pi_off, n->params + n->pi_off + (int64_t)F * A,
                      n->params + n->v_off, n->params + n->v_off + F, n->ws + n->off_logits, n->ws + n->off_value, st);
}

constexpr int kMaxNormPartials = 16384;

static int layer_wgrad(xt_net* n, int l, bool first, const void* obs, const int32_t* idx, int B, hipStream_t st) {
  Layer& L = n->layers[l];
  const void* x = first ? obs : (const void*)(n->ws + n->layers[l - 1].act_off);
  return launch_wgrad(&L.g, first ? &n->xf : nullptr, B, x, first ? idx : nullptr, n->ws + L.dact_off,
                      n->grads + L.poff, n->ws + L.slab_off, wgrad_split(L, B), st, 0, &L.last_msplit, L.slab_cap);
}

static int heads_wgrad(xt_net* n, int B, hipStream_t st) {
  Layer& Lp = n->layers[n->t_end[0] - 1];
  Layer& Lv = n->layers[n->t_end[n->n_trunks - 1] - 1];
  return launch_heads_wgrad_partial(n->ws + Lp.act_off, n->ws + Lv.act_off, B, n->feat, n->A, n->ws + n->off_dlogits,
                                    n->ws + n->off_dvalue, n->ws + n->off_hslab_pi, n->hstride_pi,
                                    n->ws + n->off_hslab_v, n->hstride_v, &n->head_chunks, st);
}

// Backward through the trunks once d(features) sits in the last layer's dact: one launch per non-first layer
// (input gradient + weight gradient [+ the head weight gradients with the very first one]), then the first layer's
// weight gradient.  (Forking the weight-gradient kernels onto side streams inside the hipGraph measured SLOWER
// than this chain -- 14.7 vs 13.9 ms per update -- and was removed.)
struct AfterFirstBwd { int (*fn)(void*); void* arg; };      // called once, right after the first backward launch

static int trunk_backward(xt_net* n, const void* obs, const int32_t* idx, int B, hipStream_t st,
                          const AfterFirstBwd* after_first = nullptr) {
  bool heads_done = false;
  bool first_done = false;
  const bool pre_ok = tuning().finalize_ticket == 0;      // (the last-block finalize form reads one partial per block)
  for (auto& L : n->layers) L.last_npre = 0;
  for (int tr = 0; tr < n->n_trunks; ++tr)
    for (int l = n->t_end[tr] - 1; l >= n->t_begin[tr]; --l) {
      const bool first = (l == n->t_begin[tr]);
      Layer& L = n->layers[l];
      if (first) {
        if (!heads_done) { if (int rc = heads_wgrad(n, B, st)) return rc; heads_done = true; }
        if (int rc = layer_wgrad(n, l, true, obs, idx, B, st)) return rc;
        continue;
      }
      Layer& Lprev = n->layers[l - 1];
      HeadWgArgs hw;
      const HeadWgArgs* hwp = nullptr;
      if (!heads_done) {
        Layer& Lp = n->layers[n->t_end[0] - 1];
        Layer& Lv = n->layers[n->t_end[n->n_trunks - 1] - 1];
        hw.f_pi = n->ws + Lp.act_off; hw.f_v = n->ws + Lv.act_off;
        hw.dlogits = n->ws + n->off_dlogits; hw.dvalue = n->ws + n->off_dvalue;
        hw.slab_pi = n->ws + n->off_hslab_pi; hw.slab_v = n->ws + n->off_hslab_v;
        hw.stride_pi = n->hstride_pi; hw.stride_v = n->hstride_v;
        hw.B = B; hw.F = n->feat; hw.A = n->A; hw.gx = (n->feat + 63) / 64; hw.nchunk = (B + 7) / 8;
        n->head_chunks = hw.nchunk;
        hwp = &hw;
        heads_done = true;
      }
      if (int rc = launch_bwd_layer(&L.g, B, n->ws + Lprev.act_off, n->ws + L.dact_off, n->params + L.poff,
                                    Lprev.g.act, n->ws + Lprev.dact_off, n->grads + L.poff, n->ws + L.slab_off,
                                    wgrad_split(L, B), hwp, &L.last_msplit, st,
                                    Lprev.mask_valid ? reinterpret_cast<const uint32_t*>(n->ws + Lprev.mask_off) : nullptr,
                                    L.slab_cap, Lprev.z_off >= 0 ? n->ws + Lprev.z_off : nullptr,
                                    // (the last layer's entry owns the FIRST partial slots: see grads_finish)
                                    pre_ok && l == (int)n->layers.size() - 1 ? n->ws + n->off_norm : nullptr, &L.last_npre))
        return rc;
      if (!first_done && after_first) { if (int rc = after_first->fn(after_first->arg)) return rc; }
      first_done = true;
    }
  return 0;
}

// ONE kernel: every slab reduction (trunk layers + heads) + the squared-norm partials
// part 0: everything; part 1: the last trunk layer + the heads (the first gradients the backward produces, 95 % of
// PpoCnn's parameters); part 2: the remaining layers.  Parts 1 / 2 serve the overlapped data-parallel exchange.
// fin->enable == 3 asks for the fused tail (reduction + norm + clip + Adam in one launch); *fused_out tells whether it ran
// that way (it is downgraded to enable == 2 when its grid could not be resident at once)
static int grads_finish(xt_net* n, int B, const FinalizeArgs* fin, hipStream_t st, int part = 0, bool* fused_out = nullptr,
                        const DpFinish* dpf = nullptr) {
  const int F = n->feat, A = n->A;
  GradTable tab;
  tab.n = 0;
  unsigned first_bucket = 0;      // entries of part 1
  XT_REQUIRE(n->layers.size() + 3 <= 12, "xt_net: too many layers for the gradient table");
  const size_t l_last = n->layers.size() - 1;
  for (size_t li = 0; li < n->layers.size(); ++li) {
    Layer& L = n->layers[li];
    if (li == l_last) first_bucket |= 1u << tab.n;
    GradEntry& E = tab.e[tab.n++];
    E.count = (L.K + 1) * L.g.N;
    E.dst = n->grads + L.poff;
    E.pre = (li == l_last && L.last_msplit == 1) ? L.last_npre : 0;
    E.nslab = L.last_msplit;
    E.src = (L.last_msplit > 1) ? n->ws + L.slab_off : E.dst;
    E.stride = E.count;
  }
  {
    first_bucket |= 1u << tab.n;
    GradEntry& E = tab.e[tab.n++];
    E.pre = 0; E.count = F * A + A; E.dst = n->grads + n->pi_off; E.src = n->ws + n->off_hslab_pi;
    E.nslab = n->head_chunks; E.stride = n->hstride_pi;
  }
  {
    first_bucket |= 1u << tab.n;
    GradEntry& E = tab.e[tab.n++];
    E.pre = 0; E.count = F + 1; E.dst = n->grads + n->v_off; E.src = n->ws + n->off_hslab_v;
    E.nslab = n->head_chunks; E.stride = n->hstride_v;
  }
  if (n->action_type == XT_ACTION_DIAG_GAUSSIAN) {   // pi_logstd: the per-sample rows are the partial slabs
    first_bucket |= 1u << tab.n;
    GradEntry& E = tab.e[tab.n++];
    E.pre = 0; E.count = A; E.dst = n->grads + n->logstd_off; E.src = n->ws + n->off_dls;
    E.nslab = n->dls_rows; E.stride = align4(A);
  }
  const unsigned all = (1u << tab.n) - 1u;
  // (part 1 runs before the other layers' backward launches have set their slab counts: its entries own the FIRST
  // partial slots, which depend on nothing else)
  const unsigned select = part == 1 ? first_bucket : part == 2 ? (all & ~first_bucket) : 0u;
  FinalizeArgs f2;
  if (fused_out) *fused_out = false;
  if (fin && fin->enable == 3) {
    f2 = *fin;
    const int cap = grads_finish_resident_blocks();
    if (part == 0 && cap > 0 && grads_finish_fused_grid(&tab) <= cap) {
      if (fused_out) *fused_out = true;
    } else {
      f2.enable = 2;
    }
    fin = &f2;
  }
  return launch_grads_finish(&tab, n->ws + n->off_norm, kMaxNormPartials, &n->norm_blocks, fin, st, select, first_bucket, dpf);
}

// loss_acc = [sum, count, data-parallel error bits, -].  TWO 8-byte memsets, the second only for a data-parallel net: ONE
// 16-byte hipMemsetAsync captured into a hipGraph wrote garbage (a pointer-looking 64-bit word) into bytes 8..15 on every
// REPLAY on this stack (ROCm 7.0 runtime of PyTorch 2.10; measured round 6, tools/dbg_acc.py) -- the 8-byte node has been
// replayed correctly since round 1
static int clear_loss_acc(const xt_net* n, float* loss_acc, hipStream_t st) {
  XT_CHECK_HIP(hipMemsetAsync(loss_acc, 0, 2 * sizeof(float), st));
  if (n->dp_world >= 1) XT_CHECK_HIP(hipMemsetAsync(loss_acc + 2, 0, 2 * sizeof(float), st));
  return 0;
}

// ---- data-parallel step helpers (xt_net_set_dp / xt_net_set_direct)
static inline int64_t dp_tail_off(const xt_net* n) { return align4(n->P); }
static inline int64_t dp_xcount(const xt_net* n) { return n->dp_world >= 1 ? align4(n->P) + kDpTailFloats : n->P; }
// what the gradient-reduction launch of a data-parallel step does besides reducing: the tail, and -- fused direct
// exchange -- the scatter into the owners' inboxes
static int dp_finish_args(xt_net* n, DpFinish* d) {
  memset(d, 0, sizeof(*d));
  if (n->dp_world < 1) return 0;
  d->rank = n->dp_rank; d->world = n->dp_world; d->rows = n->dp_rows;
  if (n->direct) {
    if (int rc = direct_fill_finish(n->direct, dp_xcount(n), d)) return rc;
    d->grads_base = n->grads;
  } else {
    d->tail = n->grads + dp_tail_off(n);
  }
  return 0;
}
// after the local gradient is complete: exchange + squared norm of the EXCHANGED gradient (fused direct form: one small
// launch reduces this rank's slice, leaves the partials and waits for the other ranks') -> what the optimiser launch needs
struct DpApply { const float* g; const float* partial; int npartial; DpStep step; int block_cap; };
static int dp_exchange(xt_net* n, hipStream_t st, float* loss_acc, DpApply* a) {
  memset(a, 0, sizeof(*a));
  a->step.world = n->dp_world; a->step.loss_scale = n->dp_loss_scale; a->step.acc = loss_acc;
  if (n->direct && n->dp_world >= 1) {
    if (int rc = direct_fill_step(n->direct, dp_xcount(n), n->P, &a->step, &a->g, &a->partial, &a->npartial, &a->block_cap))
      return rc;
    if (int rc = launch_dp_reduce_wait(&a->step, st)) return rc;
    a->block_cap = 0;                 // (the optimiser launch does not spin)
    a->step.tail = a->g + dp_tail_off(n);
    return 0;
  }
  XT_REQUIRE(n->xchg(n->grads, dp_xcount(n), n->xchg_user, st) == 0, "xt_net: gradient exchange hook failed");
  int nb = 0;
  if (int rc = launch_sqnorm_partial(n->grads, n->P, n->ws + n->off_norm, &nb, st)) return rc;
  a->g = n->grads; a->partial = n->ws + n->off_norm; a->npartial = nb;
  if (n->dp_world >= 1) a->step.tail = n->grads + dp_tail_off(n);
  return 0;
}

// mode 0: gradient was changed after grads_finish (all-reduce) -> recompute the norm;
// mode 1: squared-norm partials of grads_finish are valid -> finalize kernel;  mode 2: already finalized in-kernel;
// mode 3: partials valid, loss/step size done by grads_finish's extra block -> Adam derives the clip factor itself
static int net_apply(xt_net* n, float lr, float b1, float b2, float eps, float clip, float gscale, int mode,
                     const LossArgs* la, hipStream_t st) {
  if (mode == 3)
    return launch_adam_clip(n->params, n->grads, n->m, n->v, n->P, b1, b2, eps, n->state, n->ws + n->off_norm,
                            n->norm_blocks, clip, gscale, st, nullptr, 0, n->io_fold);
  if (mode == 1) {
    if (int rc = launch_norm_finalize(n->ws + n->off_norm, n->norm_blocks, clip, gscale, lr, b1, b2, 1, n->state, la, st))
      return rc;
  } else if (mode == 0) {
    if (int rc = launch_global_norm(n->grads, n->P, clip, gscale, lr, b1, b2, 1, n->state, n->ws + n->off_norm, st))
      return rc;
  }
  return launch_adam(n->params, n->grads, n->m, n->v, n->P, b1, b2, eps, n->state, st);
}

// tail_overlap bit 2: Adam (mode 3: every block derives the clip factor from the partials) as two launches -- the first
// layer's parameters on the compute stream, everything else on the side stream, where the next step's first-layer
// forward overlaps it.  Element-wise, same clip factor in both: bitwise the single launch's result.
static int net_apply_split(xt_net* n, float b1, float b2, float eps, float clip, float gscale, hipStream_t st) {
  const Layer& L0 = n->layers[0];
  const int64_t c0 = (int64_t)(L0.K + 1) * L0.g.N;
  if (c0 % 4 != 0 || c0 >= n->P)
    return launch_adam_clip(n->params, n->grads, n->m, n->v, n->P, b1, b2, eps, n->state, n->ws + n->off_norm,
                            n->norm_blocks, clip, gscale, st);
  XT_CHECK_HIP(hipEventRecord(n->adam_fork, st));
  XT_CHECK_HIP(hipStreamWaitEvent(n->tail_stream, n->adam_fork, 0));
  if (int rc = launch_adam_clip(n->params + c0, n->grads + c0, n->m + c0, n->v + c0, n->P - c0, b1, b2, eps, n->state,
                                n->ws + n->off_norm, n->norm_blocks, clip, gscale, n->tail_stream))
    return rc;
  XT_CHECK_HIP(hipEventRecord(n->adam_join, n->tail_stream));
  n->adam_pending = true;
  return launch_adam_clip(n->params, n->grads, n->m, n->v, c0, b1, b2, eps, n->state, n->ws + n->off_norm,
                          n->norm_blocks, clip, gscale, st);
}

// tail_overlap bit 1: called right after the first backward launch (last trunk layer + head weight gradients): their
// slab reduction -- most of the gradient bytes -- goes to the side stream, under the remaining backward launches
struct TailFork { xt_net* n; int B; hipStream_t st; bool done; };
static int tail_fork_first_bucket(void* arg) {
  TailFork* o = static_cast<TailFork*>(arg);
  xt_net* n = o->n;
  XT_CHECK_HIP(hipEventRecord(n->tail_fork, o->st));
  XT_CHECK_HIP(hipStreamWaitEvent(n->tail_stream, n->tail_fork, 0));
  if (int rc = grads_finish(n, o->B, nullptr, n->tail_stream, 1)) return rc;
  XT_CHECK_HIP(hipEventRecord(n->tail_join, n->tail_stream));
  o->done = true;
  return 0;
}

static int ppo_step(xt_net* n, const xt_ppo_cfg* c, const void* obs, const int32_t* idx, int B,
                    const void* action_v, const float* old_logp, const double* adv, const float* old_v,
                    const double* target_v, int apply, float* loss_out, float* loss_acc, hipStream_t st,
                    bool defer_join = false) {
  XT_REQUIRE(n->params && n->ws, "xt_net: buffers not bound (call xt_net_bind)");
  XT_REQUIRE(B > 0 && B <= n->maxB, "xt_net_ppo_step: batch %d outside (0,%d]", B, n->maxB);
  // apply (internal): 0 gradient only (+ a loss-reduction launch), 1 full step, 2 overlapped data-parallel exchange,
  // 3 gradient + everything of the step's tail that does not depend on the EXCHANGED gradient (loss scalars, Adam step-size
  //   advance: the extra block of the gradient-reduction launch), no update -- the data-parallel step of xt_net_ppo_train
  XT_REQUIRE(apply >= 0 && apply <= 3, "xt_net_ppo_step: bad apply mode %d", apply);
  XT_REQUIRE(apply != 2 || (n->xchg && n->xchg_stream && n->xchg_fork && n->xchg_join),
             "xt_net_ppo_step: the overlapped exchange mode needs a hook installed with XT_XCHG_OVERLAP");
  const bool gauss = (n->action_type == XT_ACTION_DIAG_GAUSSIAN);
  const int32_t* action = static_cast<const int32_t*>(action_v);
  Layer& Lp0 = n->layers[n->t_end[0] - 1];
  Layer& Lv0 = n->layers[n->t_end[n->n_trunks - 1] - 1];
  bool fused_head = (!gauss && n->A <= 8 && n->feat <= 512 && Lp0.z_off < 0 && Lv0.z_off < 0);
  const int no_defer = tuning().defer_splitk ? 0 : 1;
  if (int rc = net_forward(n, obs, idx, B, false, st, fused_head && !no_defer)) return rc;
  const float inv_b = 1.f / (float)(c->global_batch > 0 ? c->global_batch : B);
  Layer& Lp = n->layers[n->t_end[0] - 1];
  Layer& Lv = n->layers[n->t_end[n->n_trunks - 1] - 1];
  const int F = n->feat, A = n->A;
  float* lo = loss_out ? loss_out : n->ws + n->off_loss;
  if (fused_head) {
    PpoHeadArgs h;
    h.f_pi = n->ws + Lp.act_off; h.f_v = n->ws + Lv.act_off;
    h.wpi = n->params + n->pi_off; h.bpi = h.wpi + (int64_t)F * A; h.wv = n->params + n->v_off; h.bv = h.wv + F;
    h.idx = idx; h.action = action; h.old_logp = old_logp; h.old_v = old_v; h.adv = adv; h.target_v = target_v;
    h.clip_ratio = c->clip_ratio; h.ent_coef = c->ent_coef; h.vf_clip = c->vf_clip; h.critic_coef = c->critic_coef;
    h.inv_b = inv_b; h.B = B; h.F = F; h.A = A; h.act_prev = Lp.g.act; h.shared = (n->n_trunks == 1);
    h.logits = n->ws + n->off_logits; h.value = n->ws + n->off_value; h.dlogits = n->ws + n->off_dlogits;
    h.dvalue = n->ws + n->off_dvalue; h.terms = n->ws + n->off_terms;
    h.df_pi = n->ws + Lp.dact_off; h.df_v = n->ws + Lv.dact_off;
    h.part_pi = h.part_v = nullptr; h.tbias_pi = h.tbias_v = nullptr; h.feat_pi_w = h.feat_v_w = nullptr;
    h.ksplit_pi = h.ksplit_v = 1; h.act_feat = Lp.g.act; h.part_stride = (long long)B * F;
    if (Lp.last_ksplit > 1) {
      h.part_pi = n->ws + Lp.part_off; h.ksplit_pi = Lp.last_ksplit; h.feat_pi_w = n->ws + Lp.act_off;
      h.tbias_pi = n->params + Lp.poff + (int64_t)Lp.K * Lp.g.N;
    }
    if (n->n_trunks == 2 && Lv.last_ksplit > 1) {
      h.part_v = n->ws + Lv.part_off; h.ksplit_v = Lv.last_ksplit; h.feat_v_w = n->ws + Lv.act_off;
      h.tbias_v = n->params + Lv.poff + (int64_t)Lv.K * Lv.g.N;
    }
    XT_REQUIRE(n->n_trunks == 1 || ((Lp.last_ksplit > 1) == (Lv.last_ksplit > 1)),
               "xt_net: pi and v trunks ended in different split-K states (unequal trunk shapes are not supported)");
    const int hrc = launch_ppo_heads_fused(h, st);
    if (hrc > 0) return hrc;
    XT_REQUIRE(hrc == 0, "xt_net: fused PPO head kernel rejected the geometry (A=%d F=%d)", A, F);
  } else {
    if (int rc = xt_heads_fwd(n->ws + Lp.act_off, n->ws + Lv.act_off, B, F, A, n->params + n->pi_off,
                              n->params + n->pi_off + (int64_t)F * A, n->params + n->v_off, n->params + n->v_off + F,
                              n->ws + n->off_logits, n->ws + n->off_value, st))
      return rc;
    if (gauss) {
      if (int rc = launch_ppo_loss_gauss(n->ws + n->off_logits, n->params + n->logstd_off, n->ws + n->off_value, B, A,
                                         idx, static_cast<const float*>(action_v), old_logp, adv, old_v, target_v,
                                         c->clip_ratio, c->ent_coef, c->vf_clip, c->critic_coef, inv_b,
                                         n->ws + n->off_dlogits, n->ws + n->off_dvalue, n->ws + n->off_dls,
                                         (int)align4(A), n->ws + n->off_terms, st))
        return rc;
      n->dls_rows = B;
    } else if (int rc = xt_ppo_loss(n->ws + n->off_logits, n->ws + n->off_value, B, A, idx, action, old_logp, adv,
                                    old_v, target_v, c->clip_ratio, c->ent_coef, c->vf_clip, c->critic_coef, inv_b,
                                    n->ws + n->off_dlogits, n->ws + n->off_dvalue, n->ws + n->off_terms, st))
      return rc;
    // (launch_heads_dfeat reads the features only for the activation derivative: the pre-activation where one is kept)
    if (int rc = launch_heads_dfeat(n->ws + (Lp.z_off >= 0 ? Lp.z_off : Lp.act_off), n->ws + (Lv.z_off >= 0 ? Lv.z_off : Lv.act_off),
                                    B, F, A, n->params + n->pi_off,
                                    n->params + n->v_off, n->ws + n->off_dlogits, n->ws + n->off_dvalue, Lp.g.act,
                                    n->ws + Lp.dact_off, n->ws + Lv.dact_off, st))
      return rc;
  }
  // data-parallel overlap (apply == 2): reduce + exchange the last trunk layer's and the heads' gradient right after
  // the first backward launch, on the side stream, while the conv backward runs
  struct Ovl { xt_net* n; int B; hipStream_t st; float ent_coef, critic_coef, inv_b; } ovl{n, B, st, c->ent_coef, c->critic_coef, inv_b};
  AfterFirstBwd af{[](void* arg) -> int {
                     Ovl* o = static_cast<Ovl*>(arg);
                     xt_net* n = o->n;
                     if (int rc = grads_finish(n, o->B, nullptr, o->st, 1)) return rc;
                     const int64_t off = n->layers.back().poff;
                     if (n->dp_world >= 1) {
                       // the data-parallel tail rides in the FIRST bucket: the per-sample loss terms exist since the head
                       // kernel, so the step's loss share can be reduced now (not added to loss_acc: the optimiser side does)
                       if (int rc = xt_ppo_loss_reduce(n->ws + n->off_terms, o->B, o->ent_coef, o->critic_coef, o->inv_b,
                                                       n->ws + n->off_loss, nullptr, o->st))
                         return rc;
                       if (int rc = launch_dp_tail_write(n->grads + dp_tail_off(n), n->dp_rank, n->dp_rows, n->ws + n->off_loss,
                                                         nullptr, 0.f, nullptr, 0.f, 0.f, 0, o->st))
                         return rc;
                     }
                     XT_CHECK_HIP(hipEventRecord(n->xchg_fork, o->st));
                     XT_CHECK_HIP(hipStreamWaitEvent(n->xchg_stream, n->xchg_fork, 0));
                     XT_REQUIRE(n->xchg(n->grads + off, dp_xcount(n) - off, n->xchg_user, n->xchg_stream) == 0,
                                "xt_net: gradient exchange hook failed (first bucket)");
                     XT_CHECK_HIP(hipEventRecord(n->xchg_join, n->xchg_stream));
                     return 0;
                   },
                   &ovl};
  const int tov = (apply == 1) ? tail_overlap_mode(n) : 0;
  TailFork tfk{n, B, st, false};
  AfterFirstBwd tf{tail_fork_first_bucket, &tfk};
  if (int rc = trunk_backward(n, obs, idx, B, st, apply == 2 ? &af : (tov & 1) ? &tf : nullptr)) return rc;
  if (apply == 2) {
    if (int rc = grads_finish(n, B, nullptr, st, 2)) return rc;
    if (n->dp_world >= 1) return 0;        // (loss share already in the tail of the first bucket)
    return xt_ppo_loss_reduce(n->ws + n->off_terms, B, c->ent_coef, c->critic_coef, inv_b, lo, loss_acc, st);
  }
  LossArgs la{};
  la.terms = n->ws + n->off_terms; la.B = B; la.ent_coef = c->ent_coef; la.critic_coef = c->critic_coef;
  la.inv_b = inv_b; la.out = lo; la.acc = loss_acc;
  if (apply == 3) {
    FinalizeArgs fin{};
    fin.enable = 2; fin.counter = reinterpret_cast<unsigned int*>(n->ws + n->off_counter);
    fin.clip_norm = c->max_grad_norm; fin.grad_scale = c->grad_scale; fin.lr = c->lr; fin.beta1 = c->beta1;
    fin.beta2 = c->beta2; fin.state = n->state; fin.loss = la;
    DpFinish dpf;
    if (int rc = dp_finish_args(n, &dpf)) return rc;
    if (n->dp_world >= 1) fin.loss.acc = nullptr;      // the GLOBAL loss is added on the optimiser side, from the exchanged tail
    return grads_finish(n, B, &fin, st, 0, nullptr, &dpf);
  }
  if (apply == 1) {
    const int tail_mode = tuning().finalize_ticket ? 1 : 2;     // 1: the old "last block finalises" form (A/B)
    FinalizeArgs fin{};
    fin.enable = tail_mode; fin.counter = reinterpret_cast<unsigned int*>(n->ws + n->off_counter);
    fin.clip_norm = c->max_grad_norm; fin.grad_scale = c->grad_scale; fin.lr = c->lr; fin.beta1 = c->beta1;
    fin.beta2 = c->beta2; fin.state = n->state; fin.loss = la;
    bool fused = false;
    if (tfk.done) {
      XT_CHECK_HIP(hipStreamWaitEvent(st, n->tail_join, 0));
      if (int rc = grads_finish(n, B, &fin, st, 2)) return rc;
    } else {
      if (tail_mode == 2 && !tov && tuning().tail_fused) {
        fin.enable = 3;
        fin.ap.params = n->params; fin.ap.m = n->m; fin.ap.v = n->v; fin.ap.grads = n->grads; fin.ap.eps = c->eps;
      }
      if (int rc = grads_finish(n, B, &fin, st, 0, &fused)) return rc;
    }
    if (fused) return 0;
    if (tov & 2) {
      if (int rc = net_apply_split(n, c->beta1, c->beta2, c->eps, c->max_grad_norm, c->grad_scale, st)) return rc;
      return defer_join ? 0 : join_pending_update(n, st);
    }
    return net_apply(n, c->lr, c->beta1, c->beta2, c->eps, c->max_grad_norm, c->grad_scale, tail_mode == 1 ? 2 : 3,
                     nullptr, st);
  }
  if (int rc = grads_finish(n, B, nullptr, st)) return rc;
  // gradient only (data parallel): still report the local loss
  return xt_ppo_loss_reduce(n->ws + n->off_terms, B, c->ent_coef, c->critic_coef, inv_b, lo, loss_acc, st);
}

// Replay the hipGraph cached under `key`, capturing `enqueue` (on the net's private stream) on a miss.
template <typename F>
static int graph_run(xt_net* net, const char* key, hipStream_t st, F enqueue) {
  xt_net::GraphSlot* slot = nullptr;
  for (auto& g : net->gslots) if (g.exec && g.key == key) slot = &g;
  if (!slot) {
    slot = &net->gslots[0];
    for (auto& g : net->gslots) {
      if (!g.exec) { slot = &g; break; }
      if (g.used < slot->used) slot = &g;
    }
    if (slot->exec) { hipGraphExecDestroy(slot->exec); slot->exec = nullptr; slot->key.clear(); }
    hipGraph_t graph = nullptr;
    if (!net->cap_stream) XT_CHECK_HIP(hipStreamCreateWithFlags(&net->cap_stream, hipStreamNonBlocking));
    hipStream_t cs = net->cap_stream;
    XT_CHECK_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
    int rc = enqueue(cs);
    hipError_t e = hipStreamEndCapture(cs, &graph);
    if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
    XT_CHECK_HIP(e);
    e = hipGraphInstantiate(&slot->exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    XT_CHECK_HIP(e);
    slot->key = key;
  }
  slot->used = ++net->gclock;
  XT_CHECK_HIP(hipGraphLaunch(slot->exec, st));
  return 0;
}

// ImpalaCnnOpt.train (impala_cnn_opt.py:251-265) on one chunk of nfr = n_traj * T frames.  lr_dev (may be null): the
// step size in device memory (lr_schedule evaluated by the caller; lets a replayed hipGraph see a new value).
static int impala_step(xt_net* n, const xt_impala_cfg* c, const void* obs, int nfr, const float* bp_logits,
                       const int32_t* action, const uint8_t* done, const float* reward, int apply, const float* lr_dev,
                       float* loss_out, float* loss_acc, hipStream_t st, bool defer_join = false, bool first_chunk = false) {
  const int T = c->sample_batch_step;
  XT_REQUIRE(T >= 2 && nfr > 0 && nfr % T == 0, "xt_net_impala_step: n=%d must be a multiple of sample_batch_step=%d",
             nfr, T);
  XT_REQUIRE(nfr <= n->maxB, "xt_net_impala_step: %d frames > max batch %d", nfr, n->maxB);
  const int ntraj = nfr / T, F = n->feat, A = n->A;
  Layer& Lp = n->layers[n->t_end[0] - 1];
  Layer& Lv = n->layers[n->t_end[n->n_trunks - 1] - 1];
  float* lo = n->ws + n->off_loss;   // [0] = loss, [4 .. 4 + n_traj) per-trajectory sums
  // fused form (ImpalaCnnOpt: one trunk, A <= 8, T <= 256): split-K finish + heads in one launch, v-trace + loss +
  // d(heads) + d(features) in the next, loss scalar in the gradient-reduction launch
  const bool fused = (n->n_trunks == 1 && A <= 8 && F <= 512 && T <= 256 && Lp.z_off < 0);
  bool loss_pending = false;
  // the first chunk of a train clears loss_acc: in the fused form by WRITING {loss, 1} where later chunks add (LossArgs.acc_set,
  // the extra block of the gradient-reduction launch) -- no memset node in front of the train (a 2.4-4.8 us fill kernel + a
  // launch boundary of a 128-frame train's 85 us) --, otherwise with the memset
  const bool set_acc = first_chunk && fused && apply == 1 && loss_acc && n->dp_world < 1;
  if (first_chunk && !set_acc && loss_acc)
    if (int rc = clear_loss_acc(n, loss_acc, st)) return rc;
  if (fused) {
    if (int rc = net_forward(n, obs, nullptr, nfr, false, st, true)) return rc;
    ImpalaHeadArgs h{};
    h.feat = n->ws + Lp.act_off; h.wpi = n->params + n->pi_off; h.bpi = h.wpi + (int64_t)F * A;
    h.wv = n->params + n->v_off; h.bv = h.wv + F; h.B = nfr; h.F = F; h.A = A;
    h.logits = n->ws + n->off_logits; h.value = n->ws + n->off_value; h.act_feat = Lp.g.act;
    h.ksplit = 1; h.part_stride = (long long)nfr * F;
    if (Lp.last_ksplit > 1) {
      h.part = n->ws + Lp.part_off; h.ksplit = Lp.last_ksplit; h.feat_w = n->ws + Lp.act_off;
      h.tbias = n->params + Lp.poff + (int64_t)Lp.K * Lp.g.N;
    }
    int rc = launch_impala_heads_fwd(h, st);
    if (rc > 0) return rc;
    XT_REQUIRE(rc == 0, "xt_net_impala_step: fused head kernel rejected the geometry (A=%d F=%d)", A, F);
    ImpalaLossArgs q{};
    q.logits = n->ws + n->off_logits; q.baseline = n->ws + n->off_value; q.bp_logits = bp_logits; q.action = action;
    q.done = done; q.reward = reward; q.T = T; q.A = A; q.F = F; q.act_prev = Lp.g.act; q.gamma = c->gamma;
    q.dlogits = n->ws + n->off_dlogits; q.dbaseline = n->ws + n->off_dvalue; q.traj_loss = lo + 4;
    q.feat = n->ws + Lp.act_off; q.wpi = n->params + n->pi_off; q.wv = n->params + n->v_off;
    q.dfeat = n->ws + Lp.dact_off;
    rc = launch_impala_vtrace_bwd(q, ntraj, st);
    if (rc > 0) return rc;
    XT_REQUIRE(rc == 0, "xt_net_impala_step: fused v-trace kernel rejected the geometry (T=%d A=%d)", T, A);
    loss_pending = true;
  } else {
    if (int rc = net_forward(n, obs, nullptr, nfr, true, st)) return rc;
    if (int rc = xt_impala_loss(n->ws + n->off_logits, n->ws + n->off_value, bp_logits, action, done, reward, ntraj, T,
                                A, c->gamma, n->ws + n->off_dlogits, n->ws + n->off_dvalue, lo,
                                (apply == 3 && n->dp_world >= 1) ? nullptr : loss_acc,      // (tail mode: the optimiser side adds the GLOBAL loss)
                                nullptr, nullptr, st))
      return rc;
    if (loss_out) XT_CHECK_HIP(hipMemcpyAsync(loss_out, lo, sizeof(float), hipMemcpyDeviceToDevice, st));
    if (int rc = launch_heads_dfeat(n->ws + (Lp.z_off >= 0 ? Lp.z_off : Lp.act_off), n->ws + (Lv.z_off >= 0 ? Lv.z_off : Lv.act_off), nfr, F, A, n->params + n->pi_off,
                                    n->params + n->v_off, n->ws + n->off_dlogits, n->ws + n->off_dvalue, Lp.g.act,
                                    n->ws + Lp.dact_off, n->ws + Lv.dact_off, st))
      return rc;
  }
  const int tov = (apply == 1 && c->opt_type == XT_OPT_ADAM) ? tail_overlap_mode(n) : 0;
  TailFork tfk{n, nfr, st, false};
  AfterFirstBwd tf{tail_fork_first_bucket, &tfk};
  if (int rc = trunk_backward(n, obs, nullptr, nfr, st, (tov & 1) ? &tf : nullptr)) return rc;
  if (!apply) {
    if (int rc = grads_finish(n, nfr, nullptr, st)) return rc;
    if (loss_pending) return launch_impala_loss_reduce(lo + 4, ntraj, loss_out ? loss_out : lo, loss_acc, st);
    return 0;
  }
  // step size bookkeeping (and the fused form's loss scalar) in an extra grads_finish block, clip factor inside the
  // Adam kernel: no finalize launch
  FinalizeArgs fin{};
  fin.enable = 2; fin.counter = nullptr; fin.clip_norm = c->grad_norm_clip; fin.grad_scale = c->grad_scale;
  fin.lr = c->lr; fin.beta1 = c->beta1; fin.beta2 = c->beta2; fin.state = n->state; fin.lr_dev = lr_dev;
  if (loss_pending) {
    fin.loss.traj_loss = lo + 4; fin.loss.n_traj = ntraj; fin.loss.out = loss_out ? loss_out : lo; fin.loss.acc = loss_acc;
    fin.loss.acc_set = set_acc ? 1 : 0;
  }
  if (apply == 3) {
    // the data-parallel chunk of xt_net_impala_train: gradient + loss scalar + step-size advance (+ the tail / the scatter
    // into the owners' inboxes), no update -- the optimiser runs on the EXCHANGED gradient
    if (!loss_pending) {      // (unfused heads: the loss scalar sits in lo[0] already; as a one-trajectory sum for the block)
      fin.loss.traj_loss = lo; fin.loss.n_traj = 1; fin.loss.out = loss_out ? loss_out : lo; fin.loss.acc = nullptr;
    }
    DpFinish dpf;
    if (int rc = dp_finish_args(n, &dpf)) return rc;
    fin.counter = reinterpret_cast<unsigned int*>(n->ws + n->off_counter);      // (the scatter ticket of the fused exchange)
    if (n->dp_world >= 1) fin.loss.acc = nullptr;
    else if (!loss_pending) fin.loss.traj_loss = nullptr;      // (xt_impala_loss has added it to loss_acc itself)
    return grads_finish(n, nfr, &fin, st, 0, nullptr, &dpf);
  }
  bool fused_tail = false;
  if (tfk.done) {
    XT_CHECK_HIP(hipStreamWaitEvent(st, n->tail_join, 0));
    if (int rc = grads_finish(n, nfr, &fin, st, 2)) return rc;
  } else {
    if (!tov && tuning().tail_fused && c->opt_type == XT_OPT_ADAM) {
      fin.enable = 3; fin.counter = reinterpret_cast<unsigned int*>(n->ws + n->off_counter);
      fin.ap.params = n->params; fin.ap.m = n->m; fin.ap.v = n->v; fin.ap.grads = n->grads; fin.ap.eps = c->eps;
    }
    if (int rc = grads_finish(n, nfr, &fin, st, 0, &fused_tail)) return rc;
  }
  if (fused_tail) return 0;
  if (tov & 2) {
    if (int rc = net_apply_split(n, c->beta1, c->beta2, c->eps, c->grad_norm_clip, c->grad_scale, st)) return rc;
    return defer_join ? 0 : join_pending_update(n, st);
  }
  if (c->opt_type == XT_OPT_RMSPROP_CENTERED)
    return launch_rmsprop_clip(n->params, n->grads, n->m, n->v, n->P, c->lr, c->rms_decay, c->rms_eps, n->state,
                               n->ws + n->off_norm, n->norm_blocks, c->grad_norm_clip, c->grad_scale, st, lr_dev);
  XT_REQUIRE(c->opt_type == XT_OPT_ADAM, "xt_net_impala_step: unknown opt_type %d", c->opt_type);
  return net_apply(n, c->lr, c->beta1, c->beta2, c->eps, c->grad_norm_clip, c->grad_scale, 3, nullptr, st);
}

}  // namespace xt

extern "C" {

int xt_abi_version(void) { return XT_ABI_VERSION; }
int32_t xt_last_launch_arith(void) { return xt::last_arith(); }
int xt_tuning_get(xt_tuning* out) {
  XT_REQUIRE(out, "xt_tuning_get: null argument");
  *out = xt::tuning();
  return 0;
}
// The knobs are independent 32-bit words: a launch running on another thread while they are changed sees, per knob, the
// old or the new value (aligned word stores), never a torn one; writers are serialised.  They are meant to be set once,
// before networks are created (split counts and captured hipGraphs are not revisited).
static std::mutex g_tuning_mu;
int xt_tuning_set(const xt_tuning* in) {
  XT_REQUIRE(in, "xt_tuning_set: null argument");
  XT_REQUIRE(in->conv1_waves == 4 || in->conv1_waves == 8, "xt_tuning_set: conv1_waves must be 4 or 8");
  XT_REQUIRE(in->direct_max_waves >= 1 && in->direct_max_waves <= 8, "xt_tuning_set: direct_max_waves outside [1,8]");
  XT_REQUIRE(in->reduce_z_lanes >= 1 && in->reduce_z_lanes <= 32 && (in->reduce_z_lanes & (in->reduce_z_lanes - 1)) == 0,
             "xt_tuning_set: reduce_z_lanes must be a power of two <= 32");
  XT_REQUIRE(in->fwd_split_target >= 1 && in->wgrad_split_target >= 1 && in->direct_waves >= 1 && in->bwd_fit_slots >= 0,
             "xt_tuning_set: block-count targets must be positive");
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  xt::tuning() = *in;
  return 0;
}
const char* xt_last_error(void) { return xt::g_err; }
const char* xt_build_arch(void) { return "gfx950"; }
#ifndef XT_SRC_SHA
#define XT_SRC_SHA "unknown"
#endif
const char* xt_build_sources_sha(void) { return XT_SRC_SHA; }

int xt_net_create(const xt_net_desc* d, int32_t max_batch, xt_net** out) {
  XT_REQUIRE(d && out && max_batch > 0, "xt_net_create: bad arguments");
  XT_REQUIRE(d->n_trunks == 1 || d->n_trunks == 2, "xt_net_create: n_trunks must be 1 or 2");
  XT_REQUIRE(d->n_layers >= d->n_trunks, "xt_net_create: every trunk needs at least one layer");
  xt_net* n = new xt_net();
  n->n_trunks = d->n_trunks; n->feat = d->feat; n->A = d->action_dim;
  n->pi_off = d->pi_off; n->v_off = d->v_off; n->P = d->n_params; n->xf = d->xf;
  n->in_h = d->in_h; n->in_w = d->in_w; n->in_c = d->in_c; n->maxB = max_batch;
  n->action_type = d->action_type; n->logstd_off = d->logstd_off; n->dls_rows = 0;
  if (d->action_type != XT_ACTION_CATEGORICAL &&
      (d->action_type != XT_ACTION_DIAG_GAUSSIAN || d->logstd_off < 0 || d->logstd_off % 4 != 0 ||
       d->logstd_off + d->action_dim > d->n_params)) {
    delete n;
    XT_REQUIRE(false, "xt_net_create: bad action_type / logstd_off");
  }
  int64_t off = 0;
  int64_t max_partial = 4;
  int cur = -1;
  for (int i = 0; i < d->n_layers; ++i) {
    xt::Layer L;
    L.g = d->layers[i].g; L.poff = d->layers[i].param_off; L.trunk = d->layers[i].trunk;
    L.K = L.g.KH * L.g.KW * L.g.C; L.OHOW = L.g.OH * L.g.OW;
    if (L.poff % 4 != 0 || L.trunk < cur || L.trunk >= d->n_trunks) {
      delete n;
      XT_REQUIRE(false, "xt_net_create: layer %d: param_off must be a multiple of 4 and trunks ordered", i);
    }
    if (L.trunk != cur) { cur = L.trunk; n->t_begin[cur] = i; }
    n->t_end[cur] = i + 1;
    const int64_t asz = xt::align4((int64_t)max_batch * L.OHOW * L.g.N);
    L.act_off = off; off += asz;
    L.dact_off = off; off += asz;
    L.z_off = -1;
    if (xt::act_needs_preact(L.g.act)) { L.z_off = off; off += asz; }
    n->layers.push_back(L);
    // wgrad slab bound: msplit <= max(1, 512/tiles) slabs of (K+1)*N floats
    const int tiles = (L.g.N <= 32) ? ((L.K + 127) / 128) : ((L.K + 63) / 64) * ((L.g.N + 63) / 64);
    int nsl = (512 / tiles) < 1 ? 1 : (512 / tiles);
    if (i == n->t_begin[cur] && nsl < max_batch) nsl = max_batch;   // first layer: one slab per sample (bf16x3 wgrad)
    if (L.g.S == 2 && L.g.KH == 4 && L.g.KW == 4 && L.g.C == 16 && L.g.N == 32 && max_batch >= 512 && nsl < 512)
      nsl = 512;                                                    // fused per-sample backward: one slab per workgroup
    if (i == n->t_begin[cur]) {                                     // ... or one per 256-position range (flattened forms)
      const int64_t nr = ((int64_t)max_batch * L.OHOW + 255) / 256;
      if (nr < 4096 && nsl < (int)nr) nsl = (int)nr;
    }
    const int64_t sl = (int64_t)nsl * (int64_t)(L.K + 1) * L.g.N;
    n->layers.back().slab_cap = nsl;
    n->layers.back().slab_off = off; off += xt::align4(sl);
    n->layers.back().last_msplit = 1;
    n->layers.back().last_ksplit = 1;
    n->layers.back().part_off = -1;
    n->layers.back().mask_off = -1;
    n->layers.back().mask_valid = 0;
    if (i == n->t_begin[cur] && L.g.N == 32 && L.g.act == XT_ACT_RELU) {   // first-layer relu sign mask (see xt_conv1.hip)
      n->layers.back().mask_off = off; off += xt::align4((int64_t)max_batch * L.OHOW);
    }
  }
  for (int tr = 0; tr < d->n_trunks; ++tr) {   // deferred split-K partials of each trunk's last layer
    xt::Layer& L = n->layers[n->t_end[tr] - 1];
    L.part_off = off; off += (int64_t)512 * 4096;
  }
  if (d->n_trunks == 2 && n->t_begin[1] == 0) {
    delete n;
    XT_REQUIRE(false, "xt_net_create: trunk 1 has no layers");
  }
  // fwd split-K partial bound: ksplit*tiles <= 512 and a tile holds 4096 outputs
  max_partial = (int64_t)512 * 4096;
  n->partial_floats = xt::align4(max_partial);
  n->off_partial = off; off += n->partial_floats;
  {
    const int chunks = (max_batch + 7) / 8;
    n->hstride_pi = xt::align4((int64_t)n->feat * n->A + n->A);
    n->hstride_v = xt::align4((int64_t)n->feat + 1);
    n->off_hslab_pi = off; off += n->hstride_pi * chunks;
    n->off_hslab_v = off; off += n->hstride_v * chunks;
  }
  n->off_logits = off; off += xt::align4((int64_t)max_batch * n->A);
  n->off_value = off; off += xt::align4(max_batch);
  n->off_dlogits = off; off += xt::align4((int64_t)max_batch * n->A);
  n->off_dvalue = off; off += xt::align4(max_batch);
  n->off_terms = off; off += xt::align4((int64_t)max_batch * 4);
  n->off_dls = off; off += (int64_t)max_batch * xt::align4(n->A);
  n->off_loss = off; off += xt::align4(8 + 2 * max_batch);     // 4 + n_traj (v-trace) / 4 + 2 B (Keras loss terms)
  n->off_norm = off; off += xt::kMaxNormPartials;
  n->off_counter = off; off += 32 * 66;   // 1 top + 64 sub ticket counters, one 128-B line each
  n->off_iofwd = off; off += 8;           // tail_in_graph: {destination, sequence number, block ticket} between its two kernels
  n->off_ioacc = off; off += 4;
  n->ws_floats = off;
  *out = n;
  return 0;
}

void xt_net_destroy(xt_net* net) {
  if (!net) return;
  for (auto& g : net->gslots) if (g.exec) hipGraphExecDestroy(g.exec);
  if (net->cap_stream) hipStreamDestroy(net->cap_stream);
  if (net->tail_stream) {
    hipStreamDestroy(net->tail_stream);
    for (hipEvent_t e : {net->tail_fork, net->tail_join, net->adam_fork, net->adam_join}) if (e) hipEventDestroy(e);
  }
  if (net->xchg_stream) { hipStreamDestroy(net->xchg_stream); hipEventDestroy(net->xchg_fork); hipEventDestroy(net->xchg_join); }
  if (net->io_mb) hipHostFree(net->io_mb);
  for (float* p : net->io_snap) if (p) hipFree(p);
  for (auto& h : net->io_sig) xt::sdma_signal_destroy(&h);
  delete net;
}

int64_t xt_net_workspace_bytes(const xt_net* net) { return net ? net->ws_floats * 4 : 0; }

int xt_net_bind(xt_net* n, float* params, float* grads, float* adam_m, float* adam_v, float* adam_state,
                void* workspace, int64_t workspace_bytes) {
  XT_REQUIRE(n && params && grads && workspace, "xt_net_bind: null buffer");
  XT_REQUIRE(workspace_bytes >= n->ws_floats * 4, "xt_net_bind: workspace too small (%lld < %lld bytes)",
             (long long)workspace_bytes, (long long)(n->ws_floats * 4));
  XT_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)workspace) & 15) == 0,
             "xt_net_bind: buffers must be 16-byte aligned");
  n->io_acc_clean = false;             // (a new workspace: the tail_in_graph accumulator lives in it)
  n->params = params; n->grads = grads; n->m = adam_m; n->v = adam_v; n->state = adam_state;
  n->ws = static_cast<float*>(workspace);
  XT_CHECK_HIP(hipMemset(n->ws + n->off_counter, 0, 32 * 66 * 4));   // ticket counter of grads_finish_kernel
  XT_CHECK_HIP(hipMemset(n->ws + n->off_iofwd, 0, 8 * 4));           // (the copy kernel's block ticket starts at 0)
  for (auto& g : n->gslots) if (g.exec) { hipGraphExecDestroy(g.exec); g.exec = nullptr; g.key.clear(); }
  return 0;
}

int xt_net_forward(xt_net* n, const void* obs, const int32_t* idx, int32_t B, float* logits, float* value,
                   void* stream) {
  XT_REQUIRE(n && n->params && n->ws, "xt_net_forward: buffers not bound");
  XT_REQUIRE(B > 0 && B <= n->maxB, "xt_net_forward: batch %d outside (0,%d]", B, n->maxB);
  hipStream_t st = xt::as_stream(stream);
  if (int rc = xt::net_forward(n, obs, idx, B, true, st)) return rc;
  if (logits) XT_CHECK_HIP(hipMemcpyAsync(logits, n->ws + n->off_logits, sizeof(float) * B * n->A, hipMemcpyDeviceToDevice, st));
  if (value) XT_CHECK_HIP(hipMemcpyAsync(value, n->ws + n->off_value, sizeof(float) * B, hipMemcpyDeviceToDevice, st));
  return 0;
}

int xt_net_ppo_step(xt_net* net, const xt_ppo_cfg* cfg, const void* obs, const int32_t* idx, int32_t B,
                    const void* action, const float* old_logp, const double* adv, const float* old_v,
                    const double* target_v, int32_t apply, float* loss_out, float* loss_acc, void* stream) {
  XT_REQUIRE(net && cfg, "xt_net_ppo_step: null argument");
  // `apply` is a boolean on the ABI; the internal tri-state (2 = overlapped data-parallel exchange) is only reachable
  // from xt_net_ppo_train with a hook installed
  return xt::ppo_step(net, cfg, obs, idx, B, action, old_logp, adv, old_v, target_v, apply ? 1 : 0, loss_out, loss_acc,
                      xt::as_stream(stream));
}

static int ppo_train_enqueue(xt_net* net, const xt_ppo_cfg* c, const void* obs, int32_t n, const int32_t* perm,
                             const void* action, const float* old_logp, const double* adv, const float* old_v,
                             const double* target_v, float* loss_acc, hipStream_t st) {
  if (int rc = xt::clear_loss_acc(net, loss_acc, st)) return rc;
  XT_REQUIRE(n < (1 << 24), "xt_net_ppo_train: %d rows do not fit the data-parallel tail's float slot", n);
  net->dp_rows = (float)n;
  for (int ep = 0; ep < c->num_sgd_iter; ++ep) {
    for (int start = 0; start < n; start += c->batch_size) {
      int B = (n - start) < c->batch_size ? (n - start) : c->batch_size;
      xt_ppo_cfg cc = *c;
      if (cc.global_batch > 0 && B != c->batch_size)   // short last minibatch: keep the local/global ratio
        cc.global_batch = (int)((long long)cc.global_batch * B / c->batch_size);
      const int32_t* rows = perm + (size_t)ep * n + start;
      if (c->shard_world > 1) {
        // strict data parallelism (ABI >= 9): this rank owns a balanced contiguous shard of every GLOBAL minibatch of the
        // shared permutation (xingtian_amd/parallel.py::shard_range); the loss means run over the global rows
        XT_REQUIRE(c->shard_rank >= 0 && c->shard_rank < c->shard_world, "xt_net_ppo_train: shard_rank %d outside [0,%d)",
                   c->shard_rank, c->shard_world);
        XT_REQUIRE(B >= c->shard_world, "xt_net_ppo_train: a minibatch of %d rows cannot be split over %d ranks", B,
                   c->shard_world);
        XT_REQUIRE(net->xchg, "xt_net_ppo_train: sharded minibatches need a gradient exchange (xt_net_set_rccl / "
                              "xt_net_set_grad_exchange)");
        const int base = B / c->shard_world, rem = B % c->shard_world;
        const int b0 = c->shard_rank * base + (c->shard_rank < rem ? c->shard_rank : rem);
        cc.global_batch = B;
        rows += b0;
        B = base + (c->shard_rank < rem ? 1 : 0);
      }
      if (!net->xchg) {
        if (int rc = xt::ppo_step(net, &cc, obs, rows, B, action, old_logp, adv, old_v,
                                  target_v, 1, nullptr, loss_acc, st, /*defer_join*/ true))
          return rc;
        continue;
      }
      // data parallel: local gradient -> exchange (SUM over the replicas, on this stream) -> norm of the exchanged
      // gradient, clip, Adam
      const int64_t off_a = net->layers.back().poff;      // first bucket = [off_a, P): last trunk layer + heads
      const bool overlap = (net->xchg_flags & XT_XCHG_OVERLAP) && net->n_trunks == 1 && net->layers.size() > 1 &&
                           off_a > 0 && net->pi_off > off_a && net->v_off > off_a && net->xchg_stream != nullptr && !net->direct;
      if (int rc = xt::ppo_step(net, &cc, obs, rows, B, action, old_logp, adv, old_v,
                                target_v, overlap ? 2 : 3, nullptr, loss_acc, st))
        return rc;
      if (overlap) {
        XT_REQUIRE(net->xchg(net->grads, off_a, net->xchg_user, st) == 0, "xt_net_ppo_train: gradient exchange hook failed");
        XT_CHECK_HIP(hipStreamWaitEvent(st, net->xchg_join, 0));      // first bucket's exchange has finished
        if (int rc = xt::net_apply(net, cc.lr, cc.beta1, cc.beta2, cc.eps, cc.max_grad_norm, cc.grad_scale, 0, nullptr, st))
          return rc;
        if (net->dp_world >= 1) {
          xt::DpStep ds{};
          ds.tail = net->grads + xt::dp_tail_off(net); ds.world = net->dp_world; ds.loss_scale = net->dp_loss_scale; ds.acc = loss_acc;
          if (int rc = xt::launch_dp_tail_consume(&ds, st)) return rc;
        }
        continue;
      }
      // (round 5) the loss scalars and the Adam step-size advance rode in the gradient-reduction launch (apply == 3); what
      // is left after the exchange: the squared-norm partials of the EXCHANGED gradient, then Adam, every block deriving
      // the clip factor from them itself.  (round 6) xt_net_set_dp: the tail of the exchanged buffer carries every rank's
      // rows and loss share -- the optimiser's block 0 adds the GLOBAL loss, no host collective per train; xt_net_set_direct:
      // the gradient reduction scattered straight into the owners' inboxes, dp_exchange is ONE small reduce launch that also
      // leaves the squared-norm partials, Adam reads the exchange block: three launches after the backward pass.
      xt::DpApply da;
      if (int rc = xt::dp_exchange(net, st, loss_acc, &da)) return rc;
      if (int rc = xt::launch_adam_clip(net->params, da.g, net->m, net->v, net->P, cc.beta1, cc.beta2, cc.eps, net->state,
                                        da.partial, da.npartial, cc.max_grad_norm, cc.grad_scale, st, &da.step, da.block_cap))
        return rc;
    }
  }
  return xt::join_pending_update(net, st);
}

int xt_net_ppo_train(xt_net* net, const xt_ppo_cfg* c, const void* obs, int32_t n, const int32_t* perm,
                     const void* action, const float* old_logp, const double* adv, const float* old_v,
                     const double* target_v, float* loss_acc, int32_t use_graph, void* stream) {
  XT_REQUIRE(net && c && obs && perm && loss_acc, "xt_net_ppo_train: null argument");
  XT_REQUIRE(n > 0 && c->batch_size > 0 && c->batch_size <= net->maxB && c->num_sgd_iter > 0,
             "xt_net_ppo_train: bad sizes (n=%d batch=%d max=%d)", n, c->batch_size, net->maxB);
  hipStream_t st = xt::as_stream(stream);
  (void)xt::tail_overlap_mode(net);      // (creates the side stream outside of any capture)
  if (!use_graph)
    return ppo_train_enqueue(net, c, obs, n, perm, action, old_logp, adv, old_v, target_v, loss_acc, st);
  char key[512];
  snprintf(key, sizeof(key), "P%d.%d.%d.%d.%d.%g.%p|%p|%p|%p|%d|%p|%p|%p|%p|%p|%p|%p|%g|%g|%g|%g|%g|%g|%g|%g|%g|%d|%d|%g|%d",
           net->xchg_flags, c->shard_rank, c->shard_world, net->dp_rank, net->dp_world, net->dp_loss_scale, (void*)net->direct,
           (void*)net->xchg, net->xchg_user, obs, n,
           (const void*)perm, (const void*)action, (const void*)old_logp, (const void*)adv, (const void*)old_v,
           (const void*)target_v, (void*)loss_acc, c->lr, c->beta1, c->beta2, c->eps, c->clip_ratio, c->ent_coef,
           c->vf_clip, c->critic_coef, c->max_grad_norm, c->batch_size, c->num_sgd_iter, c->grad_scale,
           c->global_batch);
  return xt::graph_run(net, key, st, [&](hipStream_t cs) {
    return ppo_train_enqueue(net, c, obs, n, perm, action, old_logp, adv, old_v, target_v, loss_acc, cs);
  });
}

int xt_net_impala_step(xt_net* n, const xt_impala_cfg* c, const void* obs, int32_t nfr, const float* bp_logits,
                       const int32_t* action, const uint8_t* done, const float* reward, int32_t apply,
                       float* loss_out, float* loss_acc, void* stream) {
  XT_REQUIRE(n && c && n->params && n->ws, "xt_net_impala_step: null argument / unbound buffers");
  return xt::impala_step(n, c, obs, nfr, bp_logits, action, done, reward, apply, nullptr, loss_out, loss_acc,
                         xt::as_stream(stream));
}

static int impala_train_enqueue(xt_net* net, const xt_impala_cfg* c, const void* obs, int32_t n, int32_t batch_size,
                                const float* bp_logits, const int32_t* action, const uint8_t* done, const float* reward,
                                const float* lr_steps, float* loss_acc, hipStream_t st, bool clear = true,
                                const xt::IoFold* fold = nullptr) {
  if (clear && net->xchg)       // (without an exchange the first chunk clears it itself: impala_step, first_chunk)
    if (int rc = xt::clear_loss_acc(net, loss_acc, st)) return rc;
  XT_REQUIRE(n < (1 << 24), "xt_net_impala_train: %d frames do not fit the data-parallel tail's float slot", n);
  net->dp_rows = (float)n;
  const size_t frame = (size_t)net->in_h * net->in_w * net->in_c * (net->xf.is_u8 ? 1 : 4);
  int chunk = 0;
  for (int lo = 0; lo < n; lo += batch_size, ++chunk) {
    const int nfr = (n - lo) < batch_size ? (n - lo) : batch_size;
    const float* lr_dev = lr_steps ? lr_steps + chunk : nullptr;
    if (!net->xchg) {
      XT_REQUIRE(c->shard_world <= 1, "xt_net_impala_train: sharded chunks need a gradient exchange (xt_net_set_rccl / "
                                      "xt_net_set_grad_exchange)");
      const void* o = static_cast<const char*>(obs) + frame * lo;
      net->io_fold = (lo + batch_size >= n) ? fold : nullptr;       // (the last chunk's Adam kernel carries the folded tail)
      const int rc = xt::impala_step(net, c, o, nfr, bp_logits + (size_t)lo * net->A, action + lo, done + lo, reward + lo, 1,
                                     lr_dev, nullptr, loss_acc, st, /*defer_join*/ true, /*first_chunk*/ clear && lo == 0);
      net->io_fold = nullptr;
      if (rc) return rc;
      continue;
    }
    // data parallel: local gradient of this rank's trajectories -> SUM over the replicas (the loss is a sum:
    // grad_scale = 1) -> norm of the exchanged gradient, clip, optimiser
    int lo_s = lo, nfr_s = nfr;
    if (c->shard_world > 1) {
      // strict sharding (ABI >= 10): whole-trajectory shard of this chunk (xingtian_amd/parallel.py::shard_range)
      XT_REQUIRE(c->shard_rank >= 0 && c->shard_rank < c->shard_world, "xt_net_impala_train: shard_rank %d outside [0,%d)",
                 c->shard_rank, c->shard_world);
      const int T = c->sample_batch_step, ntraj = nfr / T;
      const int base = ntraj / c->shard_world, rem = ntraj % c->shard_world;
      const int b0 = c->shard_rank * base + (c->shard_rank < rem ? c->shard_rank : rem);
      lo_s = lo + b0 * T;
      nfr_s = (base + (c->shard_rank < rem ? 1 : 0)) * T;
    }
    if (nfr_s > 0) {
      // gradient of this rank's trajectories + the loss scalar + the step-size advance (lr_schedule's value is read on the
      // device) in the gradient-reduction launch; with xt_net_set_dp also the tail, with xt_net_set_direct the scatter
      const void* o = static_cast<const char*>(obs) + frame * lo_s;
      if (int rc = xt::impala_step(net, c, o, nfr_s, bp_logits + (size_t)lo_s * net->A, action + lo_s, done + lo_s,
                                   reward + lo_s, 3, lr_dev, nullptr, loss_acc, st))
        return rc;
    } else {
      // empty shard (fewer trajectories than ranks): a ZERO contribution -- zeros, the step-size advance, the tail (loss 0)
      const int64_t xc = xt::dp_xcount(net);
      XT_CHECK_HIP(hipMemsetAsync(net->grads, 0, sizeof(float) * (size_t)xc, st));
      if (int rc = xt::launch_dp_tail_write(net->dp_world >= 1 ? net->grads + xt::dp_tail_off(net) : net->ws + net->off_loss,
                                            net->dp_world >= 1 ? net->dp_rank : 0, net->dp_rows, nullptr, net->state, c->lr,
                                            lr_dev, c->beta1, c->beta2, 1, st))
        return rc;
      if (net->direct && net->dp_world >= 1)
        if (int rc = xt::direct_launch_scatter(net->direct, net->grads, xc, st)) return rc;
    }
    // exchange -> squared norm of the EXCHANGED gradient -> the configured optimiser (Adam, or centred RMSProp), every block
    // deriving the clip factor from the partials itself
    xt::DpApply da;
    if (int rc = xt::dp_exchange(net, st, loss_acc, &da)) return rc;
    if (c->opt_type == XT_OPT_RMSPROP_CENTERED) {
      if (int rc = xt::launch_rmsprop_clip(net->params, da.g, net->m, net->v, net->P, c->lr, c->rms_decay, c->rms_eps,
                                           net->state, da.partial, da.npartial, c->grad_norm_clip, c->grad_scale, st,
                                           lr_dev, &da.step, da.block_cap))
        return rc;
    } else {
      XT_REQUIRE(c->opt_type == XT_OPT_ADAM, "xt_net_impala_train: unknown opt_type %d", c->opt_type);
      if (int rc = xt::launch_adam_clip(net->params, da.g, net->m, net->v, net->P, c->beta1, c->beta2, c->eps, net->state,
                                        da.partial, da.npartial, c->grad_norm_clip, c->grad_scale, st, &da.step, da.block_cap))
        return rc;
    }
  }
  return xt::join_pending_update(net, st);
}

namespace xt {
// device -> page-locked HOST copies as plain kernels on the learner's stream.  hipMemcpyAsync into registered host memory runs
// a runtime blit kernel anyway (__amd_rocclr_copyBuffer, 73 us for the 4 MB parameter block) but pays 15-25 us of
// submission latency on either side of it (rocprofv3 trace of the IMPALA loop, round 6: adam -> loss copy 22 us, loss copy ->
// parameter copy 14 us, parameter copy -> next train 17 us); a kernel launch behind a kernel costs ~2 us.
__global__ void __launch_bounds__(64) host_copy4_kernel(float* __restrict__ dst_host, const float* __restrict__ src) {
  if (threadIdx.x < 4) dst_host[threadIdx.x] = src[threadIdx.x];
}
// SYSTEM-scope write-through stores (sc0 sc1): whatever memory type the page-locked destination was mapped with, a store that
// has been acknowledged (s_waitcnt vmcnt(0)) is at the system's coherence point -- what lets io_publish_kernel report its own
// completion from inside the kernel without a system-scope fence (which would write the whole L2 back first)
__device__ __forceinline__ void publish_copy(float* __restrict__ dst_host, const float* __restrict__ src, long long count) {
  const long long n4 = count >> 2;
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst_host, 0, 0x7fffffff, 0x00020000);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    xt_u32x4 q;
    q.x = __float_as_uint(v.x); q.y = __float_as_uint(v.y); q.z = __float_as_uint(v.z); q.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(q, rs, (int)(i * 16), 0, /*sc0*/ 1 | kAuxSc1);
  }
#endif
  if (blockIdx.x == 0 && threadIdx.x < (count & 3))
    __hip_atomic_store(dst_host + (n4 << 2) + threadIdx.x, src[(n4 << 2) + threadIdx.x], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(256) host_publish_kernel(float* __restrict__ dst_host, const float* __restrict__ src,
                                                           long long count) {
  publish_copy(dst_host, src, count);
}
// xt_train_io.tail_in_graph: the same two copies as the LAST KERNELS OF THE TRAIN (inside its replayed hipGraph: the arguments
// are fixed, so what changes from train to train travels through the mailbox).  The first kernel reads {destination, sequence
// number} from the page-locked mailbox, hands them to the copy kernel through device memory -- the learner thread rewrites the
// mailbox for the next train as soon as it has seen this train's loss, possibly while the copy kernel is still starting -- and
// writes loss_acc + the sequence number back: data, a system-scope fence, then the word the learner thread polls.
__global__ void __launch_bounds__(64) io_loss_kernel(IoMailbox* __restrict__ mb, float* __restrict__ acc,
                                                     unsigned long long* __restrict__ fwd, float* __restrict__ loss_acc_out) {
  const int t = threadIdx.x;
  uint32_t seq = 0;
  if (t == 0) {
    const unsigned long long dst = __hip_atomic_load(&mb->publish_dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    seq = __hip_atomic_load(&mb->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    fwd[0] = dst;
    fwd[1] = seq;
  }
  if (t < 4) {
    const float v = acc[t];
    __hip_atomic_store(&mb->loss[t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    loss_acc_out[t] = v;      // (the caller's device-side loss_acc, as every other train entry point leaves it)
    acc[t] = 0.f;             // re-armed for the next train: its graph has no memset node
  }
  // system-scope stores into coherent host memory: acknowledged = visible to the host, so the data only has to be
  // acknowledged before the sequence number leaves (a system-scope FENCE would write the whole L2 back first)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (t == 0) __hip_atomic_store(&mb->loss_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... and tells the host when the block has landed: every workgroup's stores, a system-scope fence, a ticket; the last one
// writes the train's sequence number into the mailbox (no event record behind the graph: a record costs the NEXT graph on the
// stream ~20 us of start latency, rocprofv3 trace of the loop, round 6)
__global__ void __launch_bounds__(256) io_publish_kernel(unsigned long long* __restrict__ fwd, const float* __restrict__ src,
                                                         long long count, IoMailbox* __restrict__ mb) {
  float* dst = reinterpret_cast<float*>(fwd[0]);
  if (!dst) return;
  publish_copy(dst, src, count);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (this workgroup's system-scope stores are acknowledged)
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int* ticket = reinterpret_cast<unsigned int*>(fwd + 2);
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      *ticket = 0u;
      __hip_atomic_store(&mb->publish_seq, (uint32_t)fwd[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// the snapshot form of the train's last kernel (tail_in_graph = 2): the new parameters -> the snapshot buffer of this train's
// parity at HBM speed, only when a destination was announced, with system-scope write-through stores -- the SDMA engine reads
// memory, not an XCD's L2 -- and the same in-kernel completion report as io_publish_kernel (mailbox word snap_seq)
__global__ void __launch_bounds__(256) io_snapshot_kernel(unsigned long long* __restrict__ fwd, const float* __restrict__ src,
                                                          float* __restrict__ snap0, float* __restrict__ snap1, long long count,
                                                          IoMailbox* __restrict__ mb) {
  if (fwd[0] == 0ull) return;
  publish_copy((fwd[1] & 1ull) ? snap1 : snap0, src, count);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int* ticket = reinterpret_cast<unsigned int*>(fwd + 2);
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      *ticket = 0u;
      __hip_atomic_store(&mb->snap_seq, (uint32_t)fwd[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// the device-side address of a page-locked host block (hipHostMalloc'ed or hipHostRegister'ed), or nullptr
static float* host_device_ptr(void* host) {
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return static_cast<float*>(d);
}
static int io_tail_enqueue(xt_net* net, float* loss_acc, int mode, hipStream_t st) {
  unsigned long long* fwd = reinterpret_cast<unsigned long long*>(net->ws + net->off_iofwd);
  hipLaunchKernelGGL(io_loss_kernel, dim3(1), dim3(64), 0, st, net->io_mb_dev, net->ws + net->off_ioacc, fwd, loss_acc);
  XT_LAUNCH_CHECK();
  if (mode == 2) {
    hipLaunchKernelGGL(io_snapshot_kernel, dim3(512), dim3(256), 0, st, fwd, net->params, net->io_snap[0], net->io_snap[1],
                       (long long)net->P, net->io_mb_dev);
  } else {
    hipLaunchKernelGGL(io_publish_kernel, dim3(256), dim3(256), 0, st, fwd, net->params, (long long)net->P, net->io_mb_dev);
  }
  XT_LAUNCH_CHECK();
  return 0;
}
}  // namespace xt

static int impala_train_run(xt_net* net, const xt_impala_cfg* c, const void* obs, int32_t n, int32_t batch_size,
                            const float* bp_logits, const int32_t* action, const uint8_t* done, const float* reward,
                            const float* lr_steps, float* loss_acc, int32_t use_graph, int io_tail, void* stream,
                            uint32_t io_seq = 0) {
  XT_REQUIRE(net && c && obs && bp_logits && action && done && reward && loss_acc, "xt_net_impala_train: null argument");
  XT_REQUIRE(net->params && net->ws, "xt_net_impala_train: buffers not bound");
  const int T = c->sample_batch_step;
  XT_REQUIRE(n > 0 && batch_size > 0 && T >= 2, "xt_net_impala_train: bad sizes (n=%d batch=%d T=%d)", n, batch_size, T);
  // impala_opt.py:90-99 slices BATCH_SIZE rows at a time and split_batches reshapes each slice to [B, T]: every chunk
  // (including the last, shorter one) must hold whole trajectories
  XT_REQUIRE(batch_size % T == 0 && n % T == 0,
             "xt_net_impala_train: n=%d and BATCH_SIZE=%d must be multiples of sample_batch_step=%d", n, batch_size, T);
  XT_REQUIRE(batch_size <= net->maxB || n <= net->maxB, "xt_net_impala_train: chunk of %d frames > max batch %d",
             batch_size < n ? batch_size : n, net->maxB);
  hipStream_t st = xt::as_stream(stream);
  (void)xt::tail_overlap_mode(net);
  // tail_in_graph = 2 with plain Adam (no gradient exchange, no split / fused tail experiment): the tail is FOLDED into the
  // Adam kernel of the last chunk -- no kernel behind the optimiser (the two tail kernels + their launch boundaries were
  // ~15 us of a 128-frame train's ~115)
  xt::IoFold fold_args{};
  const bool fold = io_tail == 2 && c->opt_type == XT_OPT_ADAM && !net->xchg && xt::tail_overlap_mode(net) == 0 &&
                    !xt::tuning().tail_fused && net->io_snap[0] && (long long)net->P * 4 < 0x7fffffffLL;
  if (fold) {
    fold_args.mb = net->io_mb_dev; fold_args.acc = net->ws + net->off_ioacc; fold_args.loss_out = loss_acc;
    fold_args.fwd = reinterpret_cast<unsigned long long*>(net->ws + net->off_iofwd);
    fold_args.snap = net->io_snap[io_seq & 1];
  }
  auto enqueue = [&](hipStream_t cs) {
    // tail_in_graph: the chunks accumulate into the library's own 4 floats, which the loss kernel re-arms -- no memset node
    if (int rc = impala_train_enqueue(net, c, obs, n, batch_size, bp_logits, action, done, reward, lr_steps,
                                      io_tail ? net->ws + net->off_ioacc : loss_acc, cs, /*clear*/ !io_tail,
                                      fold ? &fold_args : nullptr))
      return rc;
    return (io_tail && !fold) ? xt::io_tail_enqueue(net, loss_acc, io_tail, cs) : 0;
  };
  if (!use_graph) return enqueue(st);
  char key[512];
  snprintf(key, sizeof(key), "I%d.%d.%d.%d.%p|%p|%p|%p|%d|%d|%p|%p|%p|%p|%p|%p|%g|%g|%g|%g|%g|%g|%d|%g|%d|%g|%g|%p|%d",
           c->shard_rank, c->shard_world, net->dp_rank, net->dp_world, (void*)net->direct, (void*)net->xchg, net->xchg_user, obs, n, batch_size, (const void*)bp_logits, (const void*)action,
           (const void*)done, (const void*)reward, (const void*)lr_steps, (void*)loss_acc, c->lr, c->beta1, c->beta2,
           c->eps, c->grad_norm_clip, c->gamma, c->sample_batch_step, c->grad_scale, c->opt_type, c->rms_decay, c->rms_eps,
           io_tail ? (void*)(reinterpret_cast<char*>(net->io_mb_dev) + io_tail) : nullptr,
           fold ? 1 + (int)(io_seq & 1) : 0);      // (folded: the snapshot buffer of the train's parity is a kernel argument)
  return xt::graph_run(net, key, st, enqueue);
}

int xt_net_impala_train(xt_net* net, const xt_impala_cfg* c, const void* obs, int32_t n, int32_t batch_size,
                        const float* bp_logits, const int32_t* action, const uint8_t* done, const float* reward,
                        const float* lr_steps, float* loss_acc, int32_t use_graph, void* stream) {
  return impala_train_run(net, c, obs, n, batch_size, bp_logits, action, done, reward, lr_steps, loss_acc, use_graph, 0,
                          stream);
}

// tail_in_graph: wait (bounded) until the train's first tail kernel has written `seq` into the mailbox
static int io_wait_loss(xt_net* net, uint32_t seq, hipStream_t st) {
  const uint32_t* word = &net->io_mb->loss_seq;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 1;; ++spins) {
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) return 0;
    __builtin_ia32_pause();
    if ((spins & 0x3fff) != 0) continue;
    // every ~0.3 ms: is the stream still working?  (an idle or failed stream will never write the word)
    const hipError_t q = hipStreamQuery(st);
    if (q == hipErrorNotReady) { (void)hipGetLastError(); }
    else {
      if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) return 0;
      XT_CHECK_HIP(q);
      XT_REQUIRE(false, "xt_net_impala_train_io: the stream is idle but the train's tail kernel did not report (sequence %u, "
                        "mailbox holds %u)", seq, *word);
    }
    XT_REQUIRE(std::chrono::steady_clock::now() - t0 < std::chrono::seconds(30),
               "xt_net_impala_train_io: no loss from the device after 30 s (sequence %u)", seq);
  }
}

int xt_net_impala_train_io(xt_net* net, const xt_impala_cfg* c, const void* obs, int32_t n, int32_t batch_size,
                           const float* bp_logits, const int32_t* action, const uint8_t* done, const float* reward,
                           const float* lr_steps, float* loss_acc, int32_t use_graph, const xt_train_io* io, void* stream) {
  hipStream_t st = xt::as_stream(stream);
  XT_REQUIRE(net, "xt_net_impala_train_io: null net");
  const bool tail = io && io->tail_in_graph && io->wait_loss && io->loss_host;
  const int mode = !tail ? 0 : (io->tail_in_graph == 2 ? 2 : 1);    // 2: snapshot in the graph, the runtime's D2H on the side stream
  const auto t_begin = std::chrono::steady_clock::now();
  // the rollout's copies: usually long done (they ran under the previous train) -- then there is nothing to wait for and the
  // stream is spared a cross-stream barrier in front of the train
  if (io && io->wait_event && hipEventQuery(static_cast<hipEvent_t>(io->wait_event)) != hipSuccess) {
    (void)hipGetLastError();
    XT_CHECK_HIP(hipStreamWaitEvent(st, static_cast<hipEvent_t>(io->wait_event), 0));
  }
  if (io && io->wait_dma_ticket)
    XT_REQUIRE(xt_dma_wait_upto(io->wait_dma_ticket, 30000) == 0, "xt_net_impala_train_io: the rollout's copies (ticket %llu) did "
               "not land within 30 s", (unsigned long long)io->wait_dma_ticket);
  bool published = false;
  uint32_t seq = 0;
  if (tail) {
    if (!net->io_mb) {
      void *h = nullptr, *d = nullptr;
      XT_CHECK_HIP(hipHostMalloc(&h, sizeof(xt::IoMailbox), hipHostMallocMapped | hipHostMallocCoherent));
      memset(h, 0, sizeof(xt::IoMailbox));
      if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); hipHostFree(h); XT_REQUIRE(false, "xt_net_impala_train_io: the mailbox is not mapped into the device's address space"); }
      net->io_mb = static_cast<xt::IoMailbox*>(h);
      net->io_mb_dev = static_cast<xt::IoMailbox*>(d);
    }
    if (mode == 2 && !net->io_snap[0])
      for (int k = 0; k < 2; ++k)
        XT_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&net->io_snap[k]), sizeof(float) * (size_t)xt::align4(net->P)));
    float* d = io->publish_dst ? xt::host_device_ptr(io->publish_dst) : nullptr;
    published = mode == 2 ? io->publish_dst != nullptr : (d && (reinterpret_cast<uintptr_t>(d) & 15) == 0);
    if (mode == 2 && published && !d) d = io->publish_dst;      // (only "a destination exists" travels through the mailbox)
    seq = ++net->io_seq;
    if (seq == 0) seq = ++net->io_seq;       // (0 is what an untouched mailbox holds)
    net->io_mb->publish_dst = published ? reinterpret_cast<unsigned long long>(d) : 0ull;
    net->io_mb->seq = seq;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);  // the mailbox is written before the launch's doorbell
    if (mode == 2) {
      // the snapshot buffer of this parity is free once the copy of the publish that owns it has left it -- long ago,
      // normally; otherwise it is made here and now (nobody waited for that publish yet).  (Every train: the folded form
      // snapshots whether or not a destination was announced.)
      if (const uint32_t owner = net->io_snap_owner[seq & 1].load())
        if (int rc = xt_net_io_publish_wait(net, owner, -1)) return rc;
      if (published) {
        net->io_dst[seq & 1] = io->publish_dst;
        net->io_snap_owner[seq & 1].store(seq);
      }
    }
    if (!net->io_acc_clean) {                 // first use / after a rebind or a failed enqueue: zero the accumulator once
      XT_CHECK_HIP(hipMemsetAsync(net->ws + net->off_ioacc, 0, 2 * sizeof(float), st));
      XT_CHECK_HIP(hipMemsetAsync(net->ws + net->off_ioacc + 2, 0, 2 * sizeof(float), st));
    }
    net->io_acc_clean = false;
  }
  const auto t_pre = std::chrono::steady_clock::now();
  if (int rc = impala_train_run(net, c, obs, n, batch_size, bp_logits, action, done, reward, lr_steps, loss_acc, use_graph, mode,
                                stream, seq))
    return rc;
  if (!io) return 0;
  if (tail) net->io_acc_clean = true;         // (the loss kernel is enqueued: it leaves the accumulator zero)
  const auto t_launch = std::chrono::steady_clock::now();
  auto account = [&](std::chrono::steady_clock::time_point t_post) {
    const auto t_end = std::chrono::steady_clock::now();
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::micro>(b - a).count();
    };
    net->io_us[0] += us(t_begin, t_pre); net->io_us[1] += us(t_pre, t_launch);
    net->io_us[2] += us(t_launch, t_post); net->io_us[3] += us(t_post, t_end);
    net->io_calls++;
  };
  if (io->consumed_event) XT_CHECK_HIP(hipEventRecord(static_cast<hipEvent_t>(io->consumed_event), st));
  if (io->loss_host && !tail) {
    if (float* d = xt::host_device_ptr(io->loss_host)) {
      hipLaunchKernelGGL(xt::host_copy4_kernel, dim3(1), dim3(64), 0, st, d, loss_acc);
      XT_LAUNCH_CHECK();
    } else {
      XT_CHECK_HIP(hipMemcpyAsync(io->loss_host, loss_acc, 4 * sizeof(float), hipMemcpyDeviceToHost, st));
    }
    if (io->loss_event) XT_CHECK_HIP(hipEventRecord(static_cast<hipEvent_t>(io->loss_event), st));
  }
  if (io->publish_dst && mode == 2) {
    // (the copy snapshot -> publish_dst is made by whoever waits for this publish: xt_net_io_publish_wait)
    XT_REQUIRE(!io->publish_event, "xt_net_impala_train_io: tail_in_graph = 2 reports a publish through xt_net_io_publish_wait, "
                                   "not through an event");
  } else if (io->publish_dst) {
    if (!published) {
      float* d = xt::host_device_ptr(io->publish_dst);
      if (d && (reinterpret_cast<uintptr_t>(d) & 15) == 0) {
        hipLaunchKernelGGL(xt::host_publish_kernel, dim3(256), dim3(256), 0, st, d, net->params, (long long)net->P);
        XT_LAUNCH_CHECK();
      } else {
        XT_CHECK_HIP(hipMemcpyAsync(io->publish_dst, net->params, sizeof(float) * (size_t)net->P, hipMemcpyDeviceToHost, st));
      }
    }
    if (io->publish_event) XT_CHECK_HIP(hipEventRecord(static_cast<hipEvent_t>(io->publish_event), st));
  }
  const auto t_post = std::chrono::steady_clock::now();
  if (tail) {
    if (io->wait_loss != 2) {              // (2: the caller comes back for the loss with xt_net_io_wait)
      if (int rc = io_wait_loss(net, seq, st)) return rc;
      memcpy(io->loss_host, net->io_mb->loss, 4 * sizeof(float));
    }
    account(t_post);
    return 0;
  }
  if (io->loss_host && io->loss_event && io->wait_loss)
    XT_CHECK_HIP(hipEventSynchronize(static_cast<hipEvent_t>(io->loss_event)));
  account(t_post);
  return 0;
}

int xt_net_io_wait(xt_net* net, float* loss_host4, void* stream) {
  XT_REQUIRE(net && loss_host4, "xt_net_io_wait: null argument");
  XT_REQUIRE(net->io_mb && net->io_seq, "xt_net_io_wait: no train with tail_in_graph has been enqueued on this net");
  const auto t0 = std::chrono::steady_clock::now();
  if (int rc = io_wait_loss(net, net->io_seq, xt::as_stream(stream))) return rc;
  memcpy(loss_host4, net->io_mb->loss, 4 * sizeof(float));
  net->io_us[3] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

uint32_t xt_net_io_seq(const xt_net* net) { return net ? net->io_seq : 0u; }

int32_t xt_net_io_loss_ready(const xt_net* net) {
  if (!net || !net->io_mb || !net->io_seq) return 1;
  return __atomic_load_n(&net->io_mb->loss_seq, __ATOMIC_ACQUIRE) == net->io_seq ? 1 : 0;
}

int xt_net_io_publish_wait(xt_net* net, uint32_t seq, int32_t timeout_ms) {
  XT_REQUIRE(net && net->io_mb, "xt_net_io_publish_wait: no train with tail_in_graph has been enqueued on this net");
  const int b = seq & 1;
  auto landed = [&] { const uint32_t c = net->io_copied[b].load(std::memory_order_acquire); return c != 0 && (int32_t)(c - seq) >= 0; };
  if (landed()) return 0;
  // a tail_in_graph = 2 publish (it owns its parity's snapshot buffer): wait for the snapshot's report, then the SDMA copy is
  // made HERE, by the first thread that comes for it; otherwise the in-graph copy kernel's own report is awaited
  const bool snap = net->io_snap_owner[b].load() == seq;
  if (snap && timeout_ms == 0) return 1;                     // (the query form does not start the copy)
  const uint32_t* word = snap ? &net->io_mb->snap_seq : &net->io_mb->publish_seq;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 1;; ++spins) {
    if (snap && landed()) return 0;
    if ((int32_t)(__atomic_load_n(word, __ATOMIC_ACQUIRE) - seq) >= 0) {
      if (!snap) return 0;
      uint32_t claimed = net->io_copying[b].load();
      if ((int32_t)(claimed - seq) < 0 && net->io_copying[b].compare_exchange_strong(claimed, seq)) {
        const size_t bytes = sizeof(float) * (size_t)net->P;
        if (xt::sdma_copy_d2h(net->io_dst[b], net->io_snap[b], bytes, &net->io_sig[b]) != nullptr) {
          // this process cannot (no HSA runtime to be had): the runtime's synchronous copy, a blit kernel
          XT_CHECK_HIP(hipMemcpy(net->io_dst[b], net->io_snap[b], bytes, hipMemcpyDeviceToHost));
        }
        net->io_copied[b].store(seq, std::memory_order_release);
        return 0;
      }
      // (another thread is making this copy: wait for it below)
    } else if (timeout_ms == 0) {
      return 1;                                              // (query form: not yet)
    }
    __builtin_ia32_pause();
    if ((spins & 0xfff) == 0) {
      if (timeout_ms > 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) return 1;
      std::this_thread::yield();             // (a helper thread waits here: it must not keep a core from the learner)
    }
  }
}

int xt_net_io_times(xt_net* net, double* us_out4, int64_t* calls_out, int32_t reset) {
  XT_REQUIRE(net && us_out4 && calls_out, "xt_net_io_times: null argument");
  for (int i = 0; i < 4; ++i) us_out4[i] = net->io_us[i];
  *calls_out = net->io_calls;
  if (reset) { for (double& v : net->io_us) v = 0; net->io_calls = 0; }
  return 0;
}

int xt_net_keras_impala_step(xt_net* n, const void* obs, const int32_t* idx, int32_t B, const float* adv,
                             const float* onehot, const float* target_v, float ent_coef, float* loss_out,
                             float* loss_acc, void* stream) {
  XT_REQUIRE(n && n->params && n->ws && obs && adv && onehot && target_v, "xt_net_keras_impala_step: null argument / unbound buffers");
  XT_REQUIRE(B > 0 && B <= n->maxB, "xt_net_keras_impala_step: batch %d outside (0,%d]", B, n->maxB);
  XT_REQUIRE(n->action_type == XT_ACTION_CATEGORICAL, "xt_net_keras_impala_step: categorical heads only");
  hipStream_t st = xt::as_stream(stream);
  if (int rc = xt::net_forward(n, obs, idx, B, true, st)) return rc;
  float* lo = n->ws + n->off_loss;
  if (int rc = xt_keras_impala_loss(n->ws + n->off_logits, n->ws + n->off_value, B, n->A, idx, adv, onehot, target_v,
                                    ent_coef, n->ws + n->off_dlogits, n->ws + n->off_dvalue, lo, loss_acc, st))
    return rc;
  if (loss_out) XT_CHECK_HIP(hipMemcpyAsync(loss_out, lo, 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
  xt::Layer& Lp = n->layers[n->t_end[0] - 1];
  xt::Layer& Lv = n->layers[n->t_end[n->n_trunks - 1] - 1];
  if (int rc = xt::launch_heads_dfeat(n->ws + (Lp.z_off >= 0 ? Lp.z_off : Lp.act_off), n->ws + (Lv.z_off >= 0 ? Lv.z_off : Lv.act_off), B, n->feat, n->A, n->params + n->pi_off,
                                      n->params + n->v_off, n->ws + n->off_dlogits, n->ws + n->off_dvalue, Lp.g.act,
                                      n->ws + Lp.dact_off, n->ws + Lv.dact_off, st))
    return rc;
  if (int rc = xt::trunk_backward(n, obs, idx, B, st)) return rc;
  return xt::grads_finish(n, B, nullptr, st);
}

int xt_net_set_grad_exchange_ex(xt_net* net, xt_grad_exchange_fn fn, void* user, int32_t flags) {
  XT_REQUIRE(net, "xt_net_set_grad_exchange: null net");
  XT_REQUIRE((flags & ~XT_XCHG_OVERLAP) == 0, "xt_net_set_grad_exchange_ex: unknown flags 0x%x", flags);
  net->xchg = fn;
  net->xchg_user = fn ? user : nullptr;
  net->xchg_flags = fn ? flags : 0;
  if (fn && (flags & XT_XCHG_OVERLAP) && !net->xchg_stream) {
    XT_CHECK_HIP(hipStreamCreateWithFlags(&net->xchg_stream, hipStreamNonBlocking));
    XT_CHECK_HIP(hipEventCreateWithFlags(&net->xchg_fork, hipEventDisableTiming));
    XT_CHECK_HIP(hipEventCreateWithFlags(&net->xchg_join, hipEventDisableTiming));
  }
  return 0;
}

int xt_net_set_grad_exchange(xt_net* net, xt_grad_exchange_fn fn, void* user) {
  return xt_net_set_grad_exchange_ex(net, fn, user, 0);
}

int xt_net_set_dp(xt_net* net, int32_t rank, int32_t world, float loss_scale) {
  XT_REQUIRE(net, "xt_net_set_dp: null net");
  if (world <= 0) {
    XT_REQUIRE(!net->direct, "xt_net_set_dp: detach the direct exchange (xt_net_set_direct(net, NULL)) first");
    net->dp_rank = 0; net->dp_world = 0; net->dp_loss_scale = 1.f;
    return 0;
  }
  XT_REQUIRE(world <= xt::kDpMaxWorld && rank >= 0 && rank < world, "xt_net_set_dp: rank %d / world %d (max %d)", rank, world,
             xt::kDpMaxWorld);
  XT_REQUIRE(net->grads, "xt_net_set_dp: buffers not bound");
  net->dp_rank = rank; net->dp_world = world; net->dp_loss_scale = loss_scale;
  return 0;
}

int xt_net_set_direct(xt_net* net, xt_direct_comm* comm) {
  XT_REQUIRE(net, "xt_net_set_direct: null net");
  if (!comm) {
    net->direct = nullptr;
    return xt_net_set_grad_exchange_ex(net, nullptr, nullptr, 0);
  }
  XT_REQUIRE(net->dp_world >= 1, "xt_net_set_direct: call xt_net_set_dp(net, rank, world, ...) first (the fused exchange "
                                "carries the data-parallel tail)");
  xt::DpFinish probe;
  memset(&probe, 0, sizeof(probe));
  if (int rc = xt::direct_fill_finish(comm, xt::dp_xcount(net), &probe)) return rc;
  XT_REQUIRE(probe.rank == net->dp_rank && probe.world == net->dp_world,
             "xt_net_set_direct: the comm is rank %d of %d, the net was set up as rank %d of %d", probe.rank, probe.world,
             net->dp_rank, net->dp_world);
  net->direct = comm;
  return xt_net_set_grad_exchange_ex(net, xt_direct_exchange_hook, comm, 0);
}

// the exchange served by the library: ncclAllReduce(grads, grads, count, ncclFloat32, ncclSum, comm, stream) through the
// function pointer the caller resolved from the RCCL instance of its process (dlsym / ctypes)
static int rccl_exchange(float* grads, int64_t count, void* user, void* stream) {
  xt_net* net = static_cast<xt_net*>(user);
  const int rc = net->rccl_fn(grads, grads, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, net->rccl_comm, stream);
  net->rccl_calls++;
  if (rc != 0) net->rccl_last_error = rc;
  return rc;
}

int xt_net_set_rccl(xt_net* net, void* comm, xt_nccl_allreduce_fn allreduce, int32_t flags) {
  XT_REQUIRE(net, "xt_net_set_rccl: null net");
  if (!comm || !allreduce) {
    net->rccl_comm = nullptr; net->rccl_fn = nullptr;
    return xt_net_set_grad_exchange_ex(net, nullptr, nullptr, 0);
  }
  net->rccl_comm = comm; net->rccl_fn = allreduce; net->rccl_calls = 0; net->rccl_last_error = 0;
  return xt_net_set_grad_exchange_ex(net, rccl_exchange, net, flags);
}

int xt_net_rccl_status(const xt_net* net, int32_t* calls, int32_t* last_error) {
  XT_REQUIRE(net, "xt_net_rccl_status: null net");
  if (calls) *calls = net->rccl_calls;
  if (last_error) *last_error = net->rccl_last_error;
  return 0;
}

int xt_net_apply(xt_net* n, float lr, float beta1, float beta2, float eps, float clip_norm, float grad_scale,
                 void* stream) {
  XT_REQUIRE(n && n->params && n->m && n->v && n->state, "xt_net_apply: buffers not bound");
  return xt::net_apply(n, lr, beta1, beta2, eps, clip_norm, grad_scale, 0, nullptr, xt::as_stream(stream));
}

int xt_net_time_tail(xt_net* n, float lr, float clip_norm, int32_t reps, float* ms_out, void* stream) {
  XT_REQUIRE(n && n->params && n->ws && ms_out && reps > 0, "xt_net_time_tail: bad arguments");
  hipStream_t st = xt::as_stream(stream);
  auto one = [&]() -> int {
    xt::FinalizeArgs fin{};
    fin.enable = 2; fin.counter = reinterpret_cast<unsigned int*>(n->ws + n->off_counter);
    fin.clip_norm = clip_norm; fin.grad_scale = 1.f; fin.lr = lr; fin.beta1 = 0.9f; fin.beta2 = 0.999f; fin.state = n->state;
    if (n->xchg || n->dp_world >= 1) {
      xt::DpFinish dpf;
      if (int rc = xt::dp_finish_args(n, &dpf)) return rc;
      if (int rc = xt::grads_finish(n, 1, &fin, st, 0, nullptr, &dpf)) return rc;
      xt::DpApply da;
      if (int rc = xt::dp_exchange(n, st, nullptr, &da)) return rc;
      return xt::launch_adam_clip(n->params, da.g, n->m, n->v, n->P, 0.9f, 0.999f, 1e-8f, n->state, da.partial, da.npartial,
                                  clip_norm, 1.f, st, &da.step, da.block_cap);
    }
    if (int rc = xt::grads_finish(n, 1, &fin, st)) return rc;
    return xt::net_apply(n, lr, 0.9f, 0.999f, 1e-8f, clip_norm, 1.f, 3, nullptr, st);
  };
  hipEvent_t e0, e1;
  XT_CHECK_HIP(hipEventCreate(&e0));
  XT_CHECK_HIP(hipEventCreate(&e1));
  int rc = one();
  if (!rc) {
    hipEventRecord(e0, st);
    for (int i = 0; i < reps && !rc; ++i) rc = one();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / reps;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return rc;
}

int xt_net_layer_offsets(const xt_net* n, int32_t layer, int64_t* out4) {
  XT_REQUIRE(n && out4 && layer >= 0 && layer < (int)n->layers.size(), "xt_net_layer_offsets: bad arguments");
  const xt::Layer& L = n->layers[layer];
  out4[0] = L.act_off; out4[1] = L.dact_off; out4[2] = L.slab_off; out4[3] = L.slab_cap;
  return 0;
}

int xt_net_time_layer(xt_net* n, int32_t layer, int32_t which, const void* obs, const int32_t* idx, int32_t B,
                      int32_t reps, float* ms_out, void* stream) {
  XT_REQUIRE(n && n->params && n->ws && ms_out, "xt_net_time_layer: bad arguments");
  XT_REQUIRE(layer >= 0 && layer < (int)n->layers.size() && reps > 0 && B > 0 && B <= n->maxB,
             "xt_net_time_layer: bad layer/reps/batch");
  hipStream_t st = xt::as_stream(stream);
  xt::Layer& L = n->layers[layer];
  bool first = false;
  for (int tr = 0; tr < n->n_trunks; ++tr) first |= (layer == n->t_begin[tr]);
  const void* x = first ? obs : (const void*)(n->ws + n->layers[layer - 1].act_off);
  XT_REQUIRE(which == 0 || which == 1 || ((which == 2 || which == 3) && !first), "xt_net_time_layer: bad kernel selector");
  hipEvent_t e0, e1;
  XT_CHECK_HIP(hipEventCreate(&e0));
  XT_CHECK_HIP(hipEventCreate(&e1));
  int rc = 0;
  auto one = [&]() -> int {
    if (which == 0) {
      // a trunk's last layer runs as the update runs it: split-K partials left for the fused head kernel to finish
      // (timing it with its stand-alone finish launch charged the layer for a kernel the update never launches: 13.7 vs
      // 7.9 us in-graph for ImpalaCnnOpt's 11x11 layer, round 3)
      const bool defer = L.part_off >= 0 && L.z_off < 0 && xt::tuning().defer_splitk != 0;
      return xt::launch_fwd(&L.g, first ? &n->xf : nullptr, B, x, first ? idx : nullptr, n->params + L.poff,
                            n->params + L.poff + (int64_t)L.K * L.g.N, n->ws + L.act_off,
                            n->ws + (defer ? L.part_off : n->off_partial), xt::fwd_split(L, B), st,
                            defer ? &L.last_ksplit : nullptr,
                            L.mask_off >= 0 ? reinterpret_cast<uint32_t*>(n->ws + L.mask_off) : nullptr, &L.mask_valid);
    }
    if (which == 1)
      return xt::launch_wgrad(&L.g, first ? &n->xf : nullptr, B, x, first ? idx : nullptr, n->ws + L.dact_off,
                              n->grads + L.poff, n->ws + L.slab_off, xt::wgrad_split(L, B), st, 0, &L.last_msplit,
                              L.slab_cap);
    xt::Layer& Lp = n->layers[layer - 1];
    if (which == 3)   // the fused per-layer backward launch (dgrad + wgrad) used by the update loop
      return xt::launch_bwd_layer(&L.g, B, n->ws + Lp.act_off, n->ws + L.dact_off, n->params + L.poff, Lp.g.act,
                                  n->ws + Lp.dact_off, n->grads + L.poff, n->ws + L.slab_off, xt::wgrad_split(L, B),
                                  nullptr, &L.last_msplit, st,
                                  Lp.mask_valid ? reinterpret_cast<const uint32_t*>(n->ws + Lp.mask_off) : nullptr, L.slab_cap);
    return xt::launch_dgrad(&L.g, B, n->ws + L.dact_off, n->params + L.poff, n->ws + Lp.act_off, Lp.g.act,
                            n->ws + Lp.dact_off, st);
  };
  rc = one();   // warm
  if (!rc) {
    hipEventRecord(e0, st);
    for (int i = 0; i < reps && !rc; ++i) rc = one();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / reps;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return rc;
}

}  // extern "C"
