/*
 * xt_mi355x.h -- C ABI of the MI355X-native XingTian learner-update path.
 *
 * The reference (huawei-noah/xingtian) is 100 % Python and ships no native
 * interface: the learner arithmetic is TensorFlow graph ops called from
 * xt/model/ppo/ppo.py, xt/model/impala/impala_cnn_opt.py, xt/model/impala/vtrace.py
 * and numpy in xt/agent/ppo/ppo.py.  Each entry point below therefore cites the
 * reference *call site* whose arithmetic it replaces.  A reference maintainer binds
 * this library with ctypes (cffi is not installed in the target image); the stub is
 * shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; the caller (PyTorch-ROCm
 *     tensors in our host code) owns all memory, nothing is allocated or freed here
 *     except inside an xt_net handle's own small descriptor tables;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream);
 *   - every function returns 0 on success, non-zero on error; xt_last_error() then
 *     returns a thread-local, NUL-terminated description;
 *   - layouts are TensorFlow's: activations NHWC, conv kernels HWIO [kh,kw,cin,cout],
 *     dense kernels [in,out]; a layer's kernel and bias are contiguous
 *     ([K*N] floats then [N] floats) inside one flat fp32 parameter buffer.
 */
#ifndef XT_MI355X_H_
#define XT_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XT_ABI_VERSION 12

#define XT_ACT_NONE 0
#define XT_ACT_RELU 1
#define XT_ACT_TANH 2
/* ABI >= 8: the other entries of ACTIVATION_MAP (xt/model/model_utils.py:8-20) */
#define XT_ACT_SIGMOID 3
#define XT_ACT_SOFTSIGN 4
#define XT_ACT_SOFTPLUS 5
#define XT_ACT_LEAKY_RELU 6      /* tf.nn.leaky_relu, alpha = 0.2 */
#define XT_ACT_ELU 7
#define XT_ACT_SELU 8
/* not monotonic: a network layer with one of these also keeps its PRE-activation (xt_net does; the stand-alone layer
 * entry points then take the pre-activation as `x` of xt_layer_dgrad) */
#define XT_ACT_SWISH 9           /* tf.nn.swish: x * sigmoid(x)                                   */
#define XT_ACT_GELU 10           /* xt/model/tf_utils.py:157-166 (tanh form)                      */

/* ------------------------------------------------------------------ misc */
int xt_abi_version(void);
const char* xt_last_error(void);
/* name of the gfx target the device code was compiled for ("gfx950") */
const char* xt_build_arch(void);
/* sha256 (16 hex digits) over the kernel sources this binary was compiled from, embedded at build time (ABI >= 9;
 * csrc/Makefile, -DXT_SRC_SHA): profile artefacts are tagged with it, so counters measured on another build of the
 * kernels -- or a stale prebuilt library next to newer sources -- are recognised (bench.py, tools/pmc_summary.py). */
const char* xt_build_sources_sha(void);

/* --------------------------------------------------------------- tuning */
/* Process-wide kernel-selection knobs (ABI >= 6).  The defaults (xt_tuning_get on a fresh process) are the forms
 * that measured fastest on MI355X for the BASELINE.json shapes; the alternatives are kept
 * for A/B measurements (tools/) and as fallbacks.  Set them BEFORE creating networks: captured hipGraphs and the
 * split counts cached in a net are not revisited.  Every form computes the same arithmetic up to fp32 summation
 * order; `bf16x6 = 0` / `conv1_bf16x3 = 0` select plain fp32 MFMA. */
typedef struct xt_tuning {
  int32_t bf16x6;             /* 1: conv input gradients as six bf16 MFMAs per 16-deep chunk (exact split)      */
  int32_t dgrad_all_classes;  /* 1: stride-2 VALID conv: all four parity classes of a position tile per block   */
  int32_t dgrad_tile64;       /* 1: stride-1 register-direct input gradient with 64-row tiles                   */
  int32_t dgrad_halo;         /* 1: ... with the dY halo of the tile staged in LDS                              */
  int32_t bwd_own_instance;   /* 1: halo input gradient in its own kernel instance (3 workgroups per CU)        */
  int32_t bwd_fit_slots;      /* co-resident workgroup slots a fused backward launch is cut to (0 = off)        */
  int32_t conv1_bf16x3;       /* 1: uint8 first layer on the bf16 matrix cores (exact three-way split)          */
  int32_t conv1_flat;         /* 1: first layer over the flattened position range (512-position workgroups)     */
  int32_t conv1_waves;        /* 8 | 4 waves per first-layer workgroup                                          */
  int32_t fwd_two_groups;     /* 1: two wave groups per forward block when the launch cannot fill the chip      */
  int32_t direct;             /* 1: register-direct kernels (xt_direct.hip) where they measured faster          */
  int32_t direct_fwd;         /* 1: ... for forwards                                                            */
  int32_t direct_dgrad;       /* 1: ... for input gradients                                                     */
  int32_t direct_all;         /* 1: ... for every shape inside their envelope (experiments)                     */
  int32_t direct_waves;       /* resident-wave target of a register-direct launch                               */
  int32_t direct_max_waves;   /* waves per register-direct workgroup (<= 8)                                     */
  int32_t direct_tile64_tiles;/* tile count from which the register-direct input gradient uses 64-row tiles     */
  int32_t fwd_split_target;   /* split-K forward: target block count                                            */
  int32_t wgrad_split_target; /* split-M weight gradient: target block count                                    */
  int32_t reduce_z_lanes;     /* gradient reduction: slab lanes per block (power of two <= 32)                  */
  int32_t defer_splitk;       /* 1: the fused head kernel finishes the last trunk layer's split-K partials      */
  int32_t finalize_ticket;    /* 1: last-block finalize of the norm instead of the clip factor inside Adam      */
  int32_t fwd_tiled_valid;    /* 1: un-padded fp32 layers with N <= 32 run the LDS-tiled bf16x6 forward instead
                                 of the register-direct fp32 one (ABI >= 7; PpoCnn conv2: -2 us per step)       */
  int32_t wgrad_rows;         /* fused backward of a 4x4/2 32->32 VALID conv (PpoCnn conv2), ABI >= 8:
                                 4 (default): two workgroups per CU (250 VGPRs) -- input gradient with all four taps'
                                   operands in flight, LDS-tiled im2col weight gradient with four register stages,
                                   the launch cut to 512 co-resident workgroups (20.5 vs 22.4 us per launch);
                                 0: the round-2 form (three workgroups per CU, one / two stages in flight);
                                 1 / 2: weight gradient from input ROWS staged in LDS (no im2col gather) with / without
                                   the deep input-gradient prefetch (measured slower: 24 us; kept for A/B)        */
  int32_t fwd_prefetch_all;   /* 0 (default): two reduction steps in flight.  1: bf16x6 forwards with <= 8 steps
                                 per wave group issue all operand loads up front (ABI >= 8; measured +0.1 ms)     */
  int32_t bwd_deep_prefetch;  /* 1: fused backward of Dense (1x1, stride 1) layers as a two-workgroups-per-CU instance:
                                 all (<= 8) reduction steps of the input gradient in flight, four register stages in
                                 the weight gradient, split cut to 512 co-resident workgroups (ABI >= 8)           */
  int32_t fwd_four_groups;    /* 1: bf16x6 forwards with <= one block per CU and >= 4 steps per group run FOUR 4-wave groups
                                 (a quarter of the reduction range each, one LDS stage) instead of two (ABI >= 8)  */
  int32_t reduce_deep_lanes;  /* gradient reduction: entries with at least this many slabs get up to 32 slab lanes per block
                                 (0 = off; default 128: the first layer's one-slab-per-workgroup gradients)         */
  int32_t fwd_xcd_chunk;      /* 1: LDS-tiled forwards with several N tiles / k splits give every XCD a contiguous
                                 run of the (m tile, n tile, k slice) order: tiles that share operand slices share
                                 an L2 (ABI >= 8)                                                                  */
  int32_t tail_overlap;       /* single-GPU update tail as graph branches (one-trunk nets with >= 2 layers; bit mask, ABI >= 8):
                                 1: the slab reduction of the last trunk layer + heads (95 % of PpoCnn's gradient bytes) runs on a
                                    side stream right after the first backward launch, under the conv backward;
                                 2: Adam is split into [first layer] on the compute stream and [everything else] on the side
                                    stream, which the NEXT step's first-layer forward overlaps (joined before layer 2)       */
  int32_t tail_fused;         /* 1: slab reduction + global norm + clip + Adam in ONE launch: the thread that reduced a group of
                                 elements also updates it, only the squared-norm partials and the step size cross a grid
                                 barrier (all workgroups resident, checked; Adam only; ABI >= 8)                          */
  int32_t dense_wgrad_x6;     /* 1 (default): the weight gradient of Dense trunk layers and of the generic 64x64 fused-backward
                                 pair (ImpalaCnnOpt's 11x11 conv) on the bf16 matrix cores (bf16x6, both operands split when
                                 they are written to LDS) inside the fused backward launch; 0: fp32 MFMA (ABI >= 10; measured
                                 18.0 -> 16.9 us for PpoCnn's Dense backward, -1.1 % for pong_impala_speedup; the other conv
                                 layers lose);
                                 2 = experiment: additionally the halo-instance conv weight gradient (PpoCnn conv3) in a
                                 one-LDS-stage bf16x6 form -- measured +0.9 us per step, kept for A/Bs only               */
  int32_t fwd_fuse12;         /* (round 6) > 0: a network whose first two layers are ImpalaCnnOpt's 84x84 pair (uint8 8x8/4 SAME
                               * 4 -> 16, then 4x4/2 SAME 16 -> 32) runs them as ONE launch per frame stack (conv1's output stays
                               * in LDS for conv2, xt_conv1.hip: conv_u8c4_same_fwd2_kernel) when the batch has at most this many
                               * frames.  DEFAULT 0 (two launches): built, parity-green and MEASURED SLOWER -- 87.6 vs 83.7 us
                               * per 128-frame train, same box: 128 workgroups leave half the CUs idle and conv2 from LDS on fp32
                               * MFMA is 64 MFMAs x 64 cycles per wave, two waves per SIMD = 3.4 us */
} xt_tuning;
int xt_tuning_get(xt_tuning* out);
int xt_tuning_set(const xt_tuning* in);

/* ------------------------------------------------------------- geometry */
/* One Conv2D / Dense layer as an implicit GEMM  Y[M,N] = act(im2col(X)[M,K] . W[K,N] + b).
 * Dense: H=W=KH=KW=S=1, C=in features.  pad_top/pad_left are TensorFlow's
 * (SAME puts the odd cell at bottom/right, so only top/left are needed).
 * Replaces keras Conv2D/Dense built in xt/model/model_utils.py:83-97 and
 * xt/model/impala/impala_cnn_opt.py:119-141. */
typedef struct xt_conv_geom {
  int32_t H, W, C;      /* input height, width, channels (NHWC)            */
  int32_t KH, KW, S;    /* kernel height/width, stride                      */
  int32_t PT, PL;       /* zero padding before the first row / column       */
  int32_t OH, OW, N;    /* output height, width, channels                   */
  int32_t act;          /* XT_ACT_* applied to the output                   */
} xt_conv_geom;

/* how a uint8 observation becomes fp32: (x - mean) / std ; mean==0 -> x / std
 * (layer_function xt/model/model_utils.py:187-189, state_transform :192-201) */
typedef struct xt_input_xform {
  int32_t is_u8;        /* 1: `in` is uint8, 0: `in` is float32 (no transform) */
  float mean, std;
} xt_input_xform;

/* --------------------------------------------------- returns (HBM-bound) */
/* GAE(gamma, lambda), float64, one GPU thread per trajectory, bit-exact with numpy.
 * Replaces PPO.data_proc, xt/agent/ppo/ppo.py:87-104.
 *   value   [n_traj, T+1] f32   reward [n_traj, T] f64   done [n_traj, T] u8 (0/1)
 *   adv, target_value [n_traj, T] f64     old_value [n_traj, T] f32 */
int xt_gae_f64(const float* value, const double* reward, const uint8_t* done,
               double* adv, double* target_value, float* old_value,
               int32_t n_traj, int32_t T, double gamma, double lam, void* stream);

/* The same GAE for a whole rollout of RAGGED trajectories laid out back to back as rows, the form the learner-side
 * ingest holds them in (ABI >= 9): trajectory i = rows [offsets[i], offsets[i+1]) of reward / done / adv /
 * target_value; value_rows[r] = V(s_r) (float32: the column the reference ships as `old_value`); boot[i] = V of the
 * state after trajectory i's last step (value[T], xt/agent/ppo/ppo.py:90).  One workgroup per trajectory, bit-exact
 * with the reference's numpy loop (xt/agent/ppo/ppo.py:87-104; CartPole twin cartpole_ppo.py:98-115).  offsets:
 * int32 [n_traj + 1] on the device. */
int xt_gae_f64_ragged(const float* value_rows, const float* boot, const double* reward, const uint8_t* done,
                      const int32_t* offsets, double* adv, double* target_value, int32_t n_traj, double gamma,
                      double lam, void* stream);

/* Zero-pad the innermost axis: src [rows, c_src] -> dst [rows, c_dst], elements of elem_bytes = 1 (uint8) or 4
 * (float32) bytes (ABI >= 9).  The reference's get_cnn_backbone (xt/model/model_utils.py:49-80) takes any channel
 * count (examples/ant_ppo.yaml:20: [84, 84, 3]); the layer kernels read 4-channel groups, so such observations enter
 * the first layer with a zero plane and its kernel carries a zero input-channel row that stays zero (zero operand ->
 * zero gradient -> zero Adam step): bit-for-bit the arithmetic of the unpadded layer.  fill_u8: the byte written into
 * the extra uint8 planes -- the value the input transform maps to 0 (state_mean; 0 for x / 255); float32 planes get 0. */
int xt_pad_channels(const void* src, void* dst, int64_t rows, int32_t c_src, int32_t c_dst, int32_t elem_bytes,
                    int32_t fill_u8, void* stream);

/* Advantage normalisation over the whole rollout, in place, float64 (ABI >= 8):
 *     adv <- (adv - mean(adv)) / (std(adv) + eps)        (numpy's population std)
 * the line the reference carries commented out, xt/algorithm/ppo/ppo.py:73 -- so it is an option of the plugin
 * (model_config ADV_NORM, default False), applied to the concatenated rollout before Model.train.  One workgroup,
 * wave-level (DPP shuffle) reductions in a fixed order.  stats (may be NULL): 2 doubles, mean and std. */
int xt_adv_normalize_f64(double* adv, int64_t n, double eps, double* stats, void* stream);

/* ------------------------------------------------ rollout staging (host) */
/* Copy `bytes` of an arriving rollout array from (pageable) host memory `src` into the page-locked staging buffer
 * `dst_pinned` with a pool of native worker threads (non-temporal stores), in work units of `chunk_bytes` (<= 0:
 * 256 KiB); if dev_dst != NULL, a hipMemcpyAsync to dev_dst + offset is enqueued on `stream` for every `ship_bytes`
 * (<= 0: 4 MiB) of contiguous staged data and for the remainder, so the H2D of one piece runs under the staging of
 * the next (and the H2D of one trajectory under the staging of the following one).  n_threads < 0: the tuned count
 * (xt_stage_tune); 0: the calling thread copies inline (still chunked and pipelined with the H2D).  Returns when everything is staged and enqueued; the caller releases the GIL (ctypes does).
 * Host pointers.  ABI >= 8.  Replaces the host half of the reference's rollout hand-over: np.concatenate of the
 * trajectories + the feed_dict upload of every minibatch (xt/algorithm/ppo/ppo.py:66-71, xt/model/ppo/ppo.py:123-129;
 * called from the learner thread's prepare_data loop, xt/framework/learner.py:306-313). */
int xt_stage_rows(void* dst_pinned, const void* src, int64_t bytes, void* dev_dst, int64_t chunk_bytes,
                  int64_t ship_bytes, int32_t n_threads, void* stream);
/* Measure the staging copy on this host inline and with 1/2/4/8 worker threads x {memcpy, non-temporal stores} on a
 * private sample of `sample_bytes` (staged in trajectory-sized pieces, as an ingest burst does) and keep the fastest
 * as the default of xt_stage_rows.  gbps10 (may be NULL): the ten measured rates in GB/s,
 * [memcpy: inline,1,2,4,8 threads | non-temporal: inline,1,2,4,8 threads]. */
int xt_stage_tune(int64_t sample_bytes, float* gbps10);
int xt_stage_get(int32_t* threads, int32_t* non_temporal);
int xt_stage_set(int32_t threads, int32_t non_temporal);

/* ----------------------------------------------- implicit-GEMM layer ops */
/* Forward of one layer.  in: [B,H,W,C] (u8 or f32); idx (may be NULL): [B] int32 row
 * gather -- sample b is read from in[idx[b]] (replaces the numpy fancy-index minibatch
 * gather state[0][mbinds], xt/model/ppo/ppo.py:123).  w,bias: [K,N],[N]; y: [B*OH*OW,N].
 * ksplit>1 computes partial sums into `partial` [ksplit, M, N] and finishes (bias+act)
 * with a second kernel.  Replaces the Conv2D/Dense forward inside sess.run,
 * xt/model/ppo/ppo.py:129, impala_cnn_opt.py:255. */
int xt_layer_fwd(const xt_conv_geom* g, const xt_input_xform* xf, int32_t B,
                 const void* in, const int32_t* idx, const float* w, const float* bias,
                 float* y, float* partial, int32_t ksplit, void* stream);

/* Weight+bias gradient of one layer: dW[K,N] = im2col(X)^T . dY, db[N] = sum_m dY.
 * dy is the gradient w.r.t. the PRE-activation output, [M,N].  dwb receives
 * [(K+1)*N] floats (kernel grad then bias grad).  msplit>1 reduces over M in `msplit`
 * slabs written to `slabs` [msplit,(K+1)*N] and summed by a second kernel
 * (deterministic; no atomics).  Replaces tf.gradients inside
 * AdamOptimizer.compute_gradients, xt/model/ppo/ppo.py:99. */
int xt_layer_wgrad(const xt_conv_geom* g, const xt_input_xform* xf, int32_t B,
                   const void* in, const int32_t* idx, const float* dy,
                   float* dwb, float* slabs, int32_t msplit, void* stream);

/* Input gradient of one layer, fused with the activation gradient of the producer:
 * dx[b,iy,ix,c] = act'(x)*sum dY.W^T, where x (fp32 [B,H,W,C]) is the producer's
 * post-activation output and act_prev its activation.  Strided convs are decomposed
 * into S*S parity classes so that no MAC is spent on structural zeros.
 * Limit: the gradient tensor dy (B*OH*OW*N floats) must be smaller than 2 GiB -- the kernels
 * address it with 32-bit byte offsets of buffer loads (error otherwise; the same holds for the
 * activation and gradient tensors of a layer inside xt_net_*). */
int xt_layer_dgrad(const xt_conv_geom* g, int32_t B, const float* dy, const float* w,
                   const float* x, int32_t act_prev, float* dx, void* stream);

/* --------------------------------------------------------------- heads */
/* logits[B,A] = f_pi.Wpi + bpi ; value[B] = f_v.Wv + bv.   (pi_latent / output_value
 * Dense layers, xt/model/model_utils.py:64-65; 1x1 Conv2D policy + dense baseline,
 * impala_cnn_opt.py:138-152). */
int xt_heads_fwd(const float* f_pi, const float* f_v, int32_t B, int32_t F, int32_t A,
                 const float* wpi, const float* bpi, const float* wv, const float* bv,
                 float* logits, float* value, void* stream);

/* PPO clipped-surrogate + entropy + clipped-critic loss and its gradient w.r.t.
 * logits and value.  Replaces actor_loss_with_entropy / critic_loss
 * (xt/model/ppo/__init__.py:4-25), CategoricalDist (xt/model/tf_dist.py:103-113) and
 * their tf.gradients.  Labels are gathered through idx (may be NULL) like the
 * observations.  adv/target_v are float64 as the actor produces them and are rounded
 * to fp32 on load (the reference's float32 placeholders, xt/model/ppo/ppo.py:65-68).
 * inv_b = 1/(global minibatch size).  loss_terms [B,4]: surr, entropy, vf, 0. */
int xt_ppo_loss(const float* logits, const float* value, int32_t B, int32_t A,
                const int32_t* idx, const int32_t* action, const float* old_logp,
                const double* adv, const float* old_v, const double* target_v,
                float clip_ratio, float ent_coef, float vf_clip, float critic_coef,
                float inv_b, float* dlogits, float* dvalue, float* loss_terms,
                void* stream);

/* The same loss with the continuous-action distribution: DiagGaussianDist
 * (xt/model/tf_dist.py:47-87) over dist_param = concat([pi_latent, pi_latent*0 + pi_logstd])
 * (xt/model/ppo/ppo.py:75-79; examples/pendulum_ppo.yaml).  mean [B,A] is the pi_latent
 * output, log_std [A] the trainable state-independent variable, action float32 [N,A]
 * (gathered through idx).  dmean [B,A]; dlogstd_rows [B,A] holds every sample's
 * contribution to d loss / d pi_logstd (their fixed-order sum over B is the gradient). */
int xt_ppo_loss_gauss(const float* mean, const float* log_std, const float* value, int32_t B,
                      int32_t A, const int32_t* idx, const float* action, const float* old_logp,
                      const double* adv, const float* old_v, const double* target_v,
                      float clip_ratio, float ent_coef, float vf_clip, float critic_coef,
                      float inv_b, float* dmean, float* dvalue, float* dlogstd_rows,
                      float* loss_terms, void* stream);

/* loss scalar from the per-sample terms of xt_ppo_loss:
 * out[0]=loss, out[1]=actor_loss, out[2]=critic_loss, out[3]=mean entropy;
 * if acc != NULL, acc[0] += loss (running sum for the mean over minibatches,
 * xt/model/ppo/ppo.py:130-132). */
int xt_ppo_loss_reduce(const float* loss_terms, int32_t B, float ent_coef, float critic_coef,
                       float inv_b, float* out, float* acc, void* stream);

/* IMPALA: v-trace targets + loss gradient for one chunk of n_traj trajectories of T
 * steps (flat env-major index b*T+t).  Replaces split_batches / vtrace_loss
 * (impala_cnn_opt.py:171-196,299-351) and vtrace.from_logic_outputs (vtrace.py:39-115).
 * done: u8, reward: f32 (clipped to [-1,1] here).  out: >= 4+n_traj floats, out[0]=loss (sum form).
 * vs/pg_adv (optional, may be NULL): [n_traj,T-1] for parity tests. */
int xt_impala_loss(const float* logits, const float* baseline, const float* bp_logits,
                   const int32_t* action, const uint8_t* done, const float* reward,
                   int32_t n_traj, int32_t T, int32_t A, float gamma,
                   float* dlogits, float* dbaseline, float* out, float* acc,
                   float* vs, float* pg_adv, void* stream);

/* Loss of the non-opt IMPALA models (ABI >= 5): Keras `impala_loss` on the SOFTMAX output plus 0.5 * mse of
 * the value head (xt/model/impala/impala_cnn.py:94-108 + model.compile(loss_weights), impala_mlp.py:83-93):
 *     p = softmax(logits);  L_pi = mean_{b,a}( adv_b * (-y_ba * log(p_ba + 1e-10)) - ent * (-p_ba * log(p_ba + 1e-10)) )
 *     L = L_pi + 0.5 * mean_b (value_b - target_b)^2
 * idx (may be NULL) gathers the rows of adv [N], onehot [N,A], target_v [N] that belong to this minibatch (the
 * logits / value rows are already in minibatch order).  out: >= 4 + 2*B floats; out[0] = L, out[1] = L_pi,
 * out[2] = mse, the rest is scratch.  acc (may be NULL): acc[0] += L * B, acc[1] += B (Keras reports the
 * sample-weighted mean of an epoch).  A <= 64. */
int xt_keras_impala_loss(const float* logits, const float* value, int32_t B, int32_t A, const int32_t* idx,
                         const float* adv, const float* onehot, const float* target_v, float ent_coef,
                         float* dlogits, float* dvalue, float* out, float* acc, void* stream);

/* tf.keras.optimizers.Adam(lr, clipnorm, decay) on a flat parameter buffer made of n_seg tensors
 * [seg_off[i], seg_off[i] + seg_size[i]) (ABI >= 5; impala_cnn.py:60 `Adam(lr=LR, clipnorm=40., decay=...)`):
 * every gradient TENSOR is clipped to clipnorm by its own norm (clip_by_norm; clipnorm <= 0: no clipping), then
 *     m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  theta -= lr_t * m / (sqrt(v) + eps)
 * with lr_t = lr / (1 + decay * iterations) * sqrt(1 - b2^t) / (1 - b1^t), t = iterations + 1, computed by the
 * caller (eps = 1e-7 in tf.keras).  scratch: >= 16 * n_seg floats.  n_seg <= 32. */
int xt_adam_keras(float* param, const float* grad, float* m, float* v, int32_t n_seg, const int64_t* seg_off,
                  const int64_t* seg_size, float clipnorm, float lr_t, float beta1, float beta2, float eps,
                  float* scratch, void* stream);

/* Backward through the two heads: head weight/bias gradients and the gradient w.r.t.
 * the trunk features (times the producer's activation gradient).  If f_v == f_pi the
 * trunk is shared and df_v must equal df_pi (contributions are summed). */
int xt_heads_bwd(const float* f_pi, const float* f_v, int32_t B, int32_t F, int32_t A,
                 const float* wpi, const float* wv, const float* dlogits, const float* dvalue,
                 int32_t act_prev, float* dwpi, float* dbpi, float* dwv, float* dbv,
                 float* df_pi, float* df_v, void* stream);

/* ----------------------------------------------------------- optimiser */
/* state[8] floats on the device: [0]=beta1^t [1]=beta2^t [2]=scale [3]=alpha
 * [4]=global_norm [5]=step(as float) [6]=device-side error word (0 = none; 1 = the grid barrier of the fused update tail,
 * xt_tuning.tail_fused, timed out: that update was SKIPPED, parameters untouched) [7] reserved.  Initialise with
 * xt_adam_state_init (beta powers = 1). */
int xt_adam_state_init(float* state, void* stream);

/* tf.clip_by_global_norm + tf.train.AdamOptimizer.apply_gradients
 * (xt/model/ppo/ppo.py:97-102, impala_cnn_opt.py:204-217) over one flat buffer:
 *   g  = grad * grad_scale                       (grad_scale = 1/world for a mean loss)
 *   g  = g * clip / max(||g||_2, clip)
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g*g-v)(1-b2)
 *   p -= lr_t*m/(sqrt(v)+eps)
 * scratch: >= 1024 floats. */
int xt_adam_tf_clip(float* param, const float* grad, float* m, float* v, int64_t count,
                    float lr, float beta1, float beta2, float eps, float clip_norm,
                    float grad_scale, float* state, float* scratch, void* stream);

/* the two halves of the above, for data-parallel use (all-reduce in between is the
 * caller's): xt_grad_sqnorm fills state[2..4] from the (already reduced) gradient. */
int xt_grad_global_norm(const float* grad, int64_t count, float clip_norm, float grad_scale,
                        float* state, float* scratch, void* stream);

/* ------------------------------------------------------------- network */
/* A whole actor-critic network + update loop, so that one C call enqueues a complete
 * Model.train().  Buffers stay caller-owned. */
typedef struct xt_net xt_net;

#define XT_ACTION_CATEGORICAL 0    /* action: int32 [N]       (CategoricalDist, tf_dist.py:89)  */
#define XT_ACTION_DIAG_GAUSSIAN 1  /* action: float32 [N, A]  (DiagGaussianDist, tf_dist.py:47) */

typedef struct xt_layer_desc {
  xt_conv_geom g;
  int64_t param_off;    /* offset (floats) of this layer's [K*N]+[N] block in the flat buffer */
  int32_t trunk;        /* 0 = pi (or shared) trunk, 1 = v trunk                                */
} xt_layer_desc;

typedef struct xt_net_desc {
  int32_t n_layers;
  const xt_layer_desc* layers;  /* trunk 0 layers first (in order), then trunk 1 layers */
  int32_t n_trunks;             /* 1 shared, 2 separate pi/v trunks                      */
  int32_t feat;                 /* features entering the heads                            */
  int32_t action_dim;
  int64_t pi_off, v_off;        /* offsets of [F*A]+[A] and [F]+[1] head blocks           */
  int64_t n_params;
  xt_input_xform xf;
  int32_t in_h, in_w, in_c;     /* observation shape                                      */
  int32_t action_type;          /* XT_ACTION_CATEGORICAL | XT_ACTION_DIAG_GAUSSIAN (ABI >= 2) */
  int64_t logstd_off;           /* offset of pi_logstd [A] (DiagGaussian only; multiple of 4) */
} xt_net_desc;

int xt_net_create(const xt_net_desc* desc, int32_t max_batch, xt_net** out);
void xt_net_destroy(xt_net* net);
/* bytes of fp32 workspace xt_net needs for minibatches of up to max_batch rows */
int64_t xt_net_workspace_bytes(const xt_net* net);
int xt_net_bind(xt_net* net, float* params, float* grads, float* adam_m, float* adam_v,
                float* adam_state, void* workspace, int64_t workspace_bytes);

/* forward only (Model.predict, xt/model/ppo/ppo.py:104-109): logits [B,A], value [B] */
int xt_net_forward(xt_net* net, const void* obs, const int32_t* idx, int32_t B,
                   float* logits, float* value, void* stream);

typedef struct xt_ppo_cfg {
  float lr, beta1, beta2, eps;
  float clip_ratio, ent_coef, vf_clip, critic_coef, max_grad_norm;
  int32_t batch_size, num_sgd_iter;
  float grad_scale;             /* 1/world_size for data parallel, else 1             */
  int32_t global_batch;         /* rows of the GLOBAL minibatch (for the mean); 0 -> local */
  /* ABI >= 9, xt_net_ppo_train only: shard_world > 1 = STRICT data parallelism inside the call -- every rank passes
   * the same rollout and permutations, and takes rows [b0, b1) of every global minibatch (balanced contiguous shards,
   * rank r of shard_world); the means run over the global rows (global_batch is set per minibatch), grad_scale = 1.
   * Needs a gradient exchange (xt_net_set_rccl / xt_net_set_grad_exchange).  0 / 1 = off. */
  int32_t shard_rank, shard_world;
} xt_ppo_cfg;

/* one SGD step on rows idx[0..B) of the rollout: forward, loss, backward -> net grads.
 * `apply` is a boolean: != 0 also runs clip+Adam (single GPU).  loss_out: 4 floats (see
 * xt_ppo_loss_reduce); loss_acc optional running sum. */
int xt_net_ppo_step(xt_net* net, const xt_ppo_cfg* cfg, const void* obs, const int32_t* idx,
                    int32_t B, const void* action /* int32 [N] | float32 [N,A], see XT_ACTION_* */,
                    const float* old_logp, const double* adv,
                    const float* old_v, const double* target_v, int32_t apply,
                    float* loss_out, float* loss_acc, void* stream);

/* Model.train of xt/model/ppo/ppo.py:111-132 in one call: NUM_SGD_ITER epochs x
 * ceil(n/BATCH_SIZE) minibatches; perm [num_sgd_iter, n] int32 holds the epoch
 * permutations (the reference's np.random.shuffle, injected).  loss_acc (4 floats): [0] receives the
 * SUM of minibatch losses, [1] the number of minibatches, [2] data-parallel error bits (0 = none; ABI >= 11).  use_graph != 0
 * captures the whole call into a hipGraph on first use and replays it afterwards
 * (pointers and sizes must then stay the same between calls). */
int xt_net_ppo_train(xt_net* net, const xt_ppo_cfg* cfg, const void* obs, int32_t n,
                     const int32_t* perm, const void* action, const float* old_logp,
                     const double* adv, const float* old_v, const double* target_v,
                     float* loss_acc, int32_t use_graph, void* stream);

/* Gradient exchange hook of xt_net_ppo_train (ABI >= 4): the reference's learner is single-process
 * (its grad_communicate host averaging, xt/framework/trainer.py:89-92, is dead code); data parallelism is this
 * library's own extension.  When a hook is set, every SGD step of xt_net_ppo_train runs as
 *     gradient-only step -> fn(net.grads, n_params, user, stream) -> global norm of the EXCHANGED gradient,
 *     clip, Adam
 * on the same stream, so that with use_graph != 0 the exchange is captured into the hipGraph of the update
 * together with the kernels (fn is then called at capture time only).  fn must enqueue an in-place SUM
 * all-reduce of `count` floats at `grads` on `stream` and return 0 -- e.g. a wrapper around
 * ncclAllReduce(grads, grads, count, ncclFloat, ncclSum, comm, stream) of the RCCL instance the process already
 * uses (xingtian_amd/parallel.py::RcclComm).  cfg.grad_scale / cfg.global_batch carry the 1/world scaling as in
 * the step-wise path.  fn == NULL removes the hook. */
typedef int (*xt_grad_exchange_fn)(float* grads, int64_t count, void* user, void* stream);
int xt_net_set_grad_exchange(xt_net* net, xt_grad_exchange_fn fn, void* user);
/* The same with flags (ABI >= 8).  XT_XCHG_OVERLAP (shared-trunk networks): every SGD step of xt_net_ppo_train
 * exchanges the gradient in TWO buckets -- [offset of the last trunk layer, n_params) = that layer + the heads (the
 * first gradients the backward pass produces; 95 % of PpoCnn's parameters), reduced and handed to fn on a side stream
 * of the library right after the first backward launch, so that its all-reduce runs under the conv backward; then
 * [0, that offset) on the compute stream, which finally waits for the side stream before the norm + clip + Adam.
 * fn is called twice per step, each time with its own sub-range and stream; the calls are issued in the same order on
 * every rank (RCCL's ordering requirement for one communicator). */
#define XT_XCHG_OVERLAP 1
int xt_net_set_grad_exchange_ex(xt_net* net, xt_grad_exchange_fn fn, void* user, int32_t flags);

/* The exchange served by the library itself (ABI >= 9): every hook call becomes
 *     allreduce(grads, grads, count, ncclFloat32 (7), ncclSum (0), comm, stream)
 * through the ncclAllReduce entry point the caller resolved from the RCCL instance its process already uses (dlsym /
 * ctypes on torch's librccl.so) -- no host-language trampoline on the enqueue path.  comm == NULL removes it.
 * xt_net_rccl_status: number of calls made and the last non-zero ncclResult (0 = none). */
typedef int (*xt_nccl_allreduce_fn)(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op,
                                    void* comm, void* stream);
int xt_net_set_rccl(xt_net* net, void* comm, xt_nccl_allreduce_fn allreduce, int32_t flags);
int xt_net_rccl_status(const xt_net* net, int32_t* calls, int32_t* last_error);

/* ---- Direct 2-phase all-reduce over peer-mapped device memory (ABI >= 10; SURVEY.md 8(b) `xt_allreduce_direct`).
 * Replaces: the reference's only gradient exchange, the host-side float64 sum of the whole flat gradient list through a
 * shared RawArray (xt/framework/trainer.py:86-92, message shape :139-144; dead code there) -- here between the GPUs of one
 * node, one process per GPU, without a host hop and without RCCL's ring (xGMI is a point-to-point mesh: both phases talk
 * to all N-1 peers at once).
 *
 *   xt_direct_create   allocates this rank's exchange block (flags + one inbox slot per rank + a result buffer, sized for
 *                      max_count floats) and writes its 64-byte hipIpcMemHandle_t to handle_out (may be NULL for in-process
 *                      groups).  rank in [0, world), world <= 16.  The block is UNCACHED device memory (published with
 *                      s_waitcnt + a system-scope flag store, nothing to invalidate on the consumer side); cached kinds
 *                      measured stale flag polls (plain) / wrong sums (fine-grained) across the XCDs' L2s and are not offered.
 *   xt_direct_connect  handles = world x XT_DIRECT_HANDLE_BYTES bytes, rank order (as gathered over any side channel,
 *                      e.g. torch.distributed all_gather); maps every peer's block (hipIpcOpenMemHandle).
 *   xt_direct_connect_local  the same for N logical ranks inside ONE process (ranks[q] = the comm object of rank q).
 *   xt_allreduce_direct  in-place SUM of buf[0..count) over the ranks, enqueued on `stream` as three kernels (push scatter ->
 *                      fixed-rank-order reduce + push gather -> copy back); capturable into a hipGraph (the sequence number
 *                      lives in device memory).  Every element is summed by exactly one rank in rank order 0..N-1, so all
 *                      replicas receive BIT-IDENTICAL results.  count <= max_count, buf 16-byte aligned.  All ranks must
 *                      call it the same number of times with the same count.  Waits are bounded (default 2 s): a missing
 *                      peer sets an error bit (xt_direct_status: 1 = scatter data never arrived, 2 = reduced slices never
 *                      arrived) instead of hanging the device; the bit is sticky and later waits of the comm return at once.
 *   xt_direct_exchange_hook  xt_grad_exchange_fn adapter: xt_net_set_grad_exchange_ex(net, xt_direct_exchange_hook, comm, 0)
 *                      makes xt_net_ppo_train / xt_net_impala_train exchange through this comm (one bucket per step).
 */
#define XT_DIRECT_HANDLE_BYTES 64
typedef struct xt_direct_comm xt_direct_comm;
int xt_direct_create(int32_t rank, int32_t world, int64_t max_count, void* handle_out, xt_direct_comm** out);
int xt_direct_connect(xt_direct_comm* comm, const void* handles);
int xt_direct_connect_local(xt_direct_comm* comm, xt_direct_comm* const* ranks);
int xt_allreduce_direct(xt_direct_comm* comm, float* buf, int64_t count, void* stream);
/* the all-reduce of an in-process group (xt_direct_connect_local) driven by one host thread: launches issued phase by
 * phase over the n ranks, each on its own stream */
int xt_allreduce_direct_group(int32_t n, xt_direct_comm* const* comms, float* const* bufs, int64_t count,
                              void* const* streams);
int xt_direct_exchange_hook(float* grads, int64_t count, void* user, void* stream);
int xt_direct_set_timeout_ms(xt_direct_comm* comm, int32_t ms);
/* xt_allreduce_direct as ONE launch (default, fused = 1: every block runs scatter -> reduce -> gather and only ever waits for
 * other RANKS' flags) or as the three launches described above (fused = 0; what the in-process group call always uses). */
int xt_direct_set_fused(xt_direct_comm* comm, int32_t fused);
int xt_direct_status(xt_direct_comm* comm, int32_t* calls, int32_t* seq, int32_t* error_bits);
int xt_direct_destroy(xt_direct_comm* comm);
/* ABI >= 11.  xt_direct_info: how many ranks of the group live on THIS rank's device (read from the device identity every
 * rank leaves in its exchange block; 1 on a real multi-GPU node, N when N test processes share one GPU) and the workgroup
 * cap the spinning kernels of this comm are launched with: (resident workgroups of the device) / (ranks on the device) --
 * a kernel whose blocks wait for other ranks' flags must leave room for those ranks' kernels.
 * xt_direct_read_result: the reduced buffer of the most recent exchange (tests; synchronises the device).
 * xt_direct_reset: clears the sticky error word, the tickets, the sequence number and this rank's flag words.  COLLECTIVE:
 * every rank calls it between two host barriers, with no exchange in flight (xingtian_amd/parallel.py::DirectComm.reset). */
int xt_direct_info(xt_direct_comm* comm, int32_t* ranks_on_device, int32_t* block_cap);
int xt_direct_read_result(xt_direct_comm* comm, float* host_out, int64_t count);
int xt_direct_reset(xt_direct_comm* comm);

/* ---- The data-parallel SGD step without host collectives and without gradient copies (ABI >= 11).
 * Replaces the two host-synchronous collectives per Model.train of ABI 10's learner (row-count check, global loss) and the
 * scatter / gather copies of the direct exchange; the reference's analogue of the message is xt/framework/trainer.py:139-144.
 *
 * xt_net_set_dp(net, rank, world, loss_scale): from now on the gradient buffer bound with xt_net_bind holds
 * align4(n_params) + XT_DP_TAIL_FLOATS floats and every exchange of xt_net_ppo_train / xt_net_impala_train covers the TAIL
 * too: slot [r] = rows rank r was handed for this update, slot [16 + r] = rank r's share of the step's loss, all other slots
 * zero -- after the SUM every rank holds every rank's values exactly and the optimiser kernel adds
 * loss_scale * (sum in rank order) to loss_acc[0] (PPO strict / IMPALA: 1, the shares of one sum; PPO weak: 1 / world, the
 * mean of the ranks' means) and raises loss_acc[2] (error bits, 4 = the ranks hold different numbers of rows) -- so one
 * train() is one C call and one read-back of loss_acc, as on one GPU.  world <= 0 switches it off; world == 1 is a
 * one-rank group (the whole chain runs against the rank's own buffers: how bench.py times the data-parallel form of the
 * step on one GPU).  The overlapped
 * two-bucket mode (XT_XCHG_OVERLAP) carries the tail in its first bucket.
 *
 * xt_net_set_direct(net, comm): the direct exchange FUSED into the step (needs xt_net_set_dp; comm sized for
 * align4(n_params) + XT_DP_TAIL_FLOATS floats): the gradient-reduction kernel writes every reduced float4 straight into the
 * owning peer's inbox and raises the ready flags; ONE small launch reduces this rank's slice in rank order, pushes it to every
 * peer's result buffer and leaves the squared-norm partials of the slice with it; the optimiser kernel waits for the done
 * flags, derives the clip factor from all ranks' partials (fixed order: bitwise the same factor everywhere) and reads the
 * gradient out of the result buffer.  Three launches per step tail instead of five, no scatter / gather copy, no separate
 * norm launch.  A wait that runs out (or a row mismatch) sets the sticky error word: the optimiser then SKIPS the update
 * (parameters and slots stay those of the last good step) and loss_acc[2] carries the bits to the host with the loss.
 * comm == NULL detaches. */
#define XT_DP_TAIL_FLOATS 32
int xt_net_set_dp(xt_net* net, int32_t rank, int32_t world, float loss_scale);
int xt_net_set_direct(xt_net* net, xt_direct_comm* comm);

/* One minibatch of Keras `model.fit` for the non-opt IMPALA models (ABI >= 5): forward, xt_keras_impala_loss,
 * backward; the flat gradient is left in the net's gradient buffer for xt_adam_keras.  obs rows are gathered with
 * idx like the label rows (idx may be NULL: rows 0..B-1). */
int xt_net_keras_impala_step(xt_net* net, const void* obs, const int32_t* idx, int32_t B, const float* adv,
                             const float* onehot, const float* target_v, float ent_coef, float* loss_out,
                             float* loss_acc, void* stream);

typedef struct xt_impala_cfg {
  float lr, beta1, beta2, eps;
  float grad_norm_clip, gamma;
  int32_t sample_batch_step;    /* T */
  float grad_scale;
  int32_t opt_type;             /* XT_OPT_ADAM | XT_OPT_RMSPROP_CENTERED (ABI >= 3)                          */
  float rms_decay, rms_eps;     /* tf.train.RMSPropOptimizer(LR, decay=0.99, epsilon=0.1, centered=True),    */
                                /* impala_cnn_opt.py:205-206: uses adam_m as the mean gradient `mg` and       */
                                /* adam_v as the mean square `ms` (the caller initialises ms to ONES as TF)   */
  /* strict data parallelism inside xt_net_impala_train (ABI >= 10): shard_world > 1 -> every rank is handed the SAME   */
  /* rollout and takes whole-trajectory shard shard_rank (balanced, contiguous) of every BATCH_SIZE chunk; the sum-form */
  /* loss makes the SUM of the shard gradients the chunk's gradient (grad_scale 1).  A rank whose shard of a chunk is   */
  /* empty (fewer trajectories than ranks) contributes a zero gradient.  Needs a gradient exchange.                     */
  int32_t shard_rank, shard_world;
} xt_impala_cfg;

#define XT_OPT_ADAM 0
#define XT_OPT_RMSPROP_CENTERED 1

/* ImpalaCnnOpt.train (impala_cnn_opt.py:251-265) on one chunk of n = n_traj*T frames */
int xt_net_impala_step(xt_net* net, const xt_impala_cfg* cfg, const void* obs, int32_t n,
                       const float* bp_logits, const int32_t* action, const uint8_t* done,
                       const float* reward, int32_t apply, float* loss_out, float* loss_acc,
                       void* stream);

/* IMPALAOpt.train of xt/algorithm/impala/impala_opt.py:73-106 in one call (ABI >= 6): the n frames of the
 * concatenated rollout messages are consumed in sequential chunks of batch_size frames (no shuffling; n and
 * batch_size must be multiples of sample_batch_step because split_batches, impala_cnn_opt.py:171-186, reshapes every
 * chunk to [B, T]), one ImpalaCnnOpt.train (forward, v-trace, sum-form loss, backward, clip, optimiser) per chunk.
 * lr_steps (device, may be NULL): step size of every chunk ([ceil(n/batch_size)] floats; the caller evaluates
 * lr_schedule / linear_cosine_decay, impala_cnn_opt.py:234-249, per global_step) -- read on the device so that a
 * replayed hipGraph sees new values; NULL -> cfg->lr.  loss_acc[0] receives the SUM of the chunk losses, loss_acc[1]
 * the number of chunks (the algorithm returns their mean, impala_opt.py:106).  use_graph != 0 captures the call into
 * a hipGraph on first use and replays it while pointers and sizes repeat (a small cache of graphs is kept, so that
 * alternating ingest buffer sets do not re-capture).  A gradient-exchange hook (xt_net_set_grad_exchange) makes every
 * chunk data parallel: local gradient -> exchange (SUM; the loss is a sum, grad_scale = 1) -> norm of the exchanged
 * gradient -> clip + the configured optimiser (Adam or centred RMSProp, lr_steps honoured; ABI >= 9). */
int xt_net_impala_train(xt_net* net, const xt_impala_cfg* cfg, const void* obs, int32_t n, int32_t batch_size,
                        const float* bp_logits, const int32_t* action, const uint8_t* done, const float* reward,
                        const float* lr_steps, float* loss_acc, int32_t use_graph, void* stream);

/* xt_net_impala_train plus the runtime calls a learner loop issues around it, in ONE call (ABI >= 11): the per-train host path
 * of IMPALAOpt.train (xt/algorithm/impala/impala_opt.py:73-106 inside xt/framework/learner.py:298-380) was ~10 separate
 * Python-level runtime calls (wait for the rollout's copies, launch, mark the buffer set consumed, snapshot the weights, read
 * the loss back, wait) around an 85 us train.  Every member of `io` is optional (NULL / 0 = skip):
 *   wait_event      the stream waits for this event before the first kernel (the rollout's H2D copies)
 *   consumed_event  recorded right behind the train (the rollout's buffer set may be overwritten once it has fired)
 *   loss_host       page-locked host block of 4 floats: loss_acc is copied there; loss_event is recorded behind the copy;
 *                   wait_loss != 0: the call returns when that copy has landed (hipEventSynchronize; the caller's language
 *                   runtime lock is released for the whole call by ctypes)
 *   publish_dst     page-locked HOST block of n_params floats (a transport.WeightsRing slot): the new parameters are copied
 *                   there in stream order BEHIND the loss copy (the next update cannot tear them, the caller's loss wait does
 *                   not include them); publish_event is recorded behind the copy (the ring's committer thread waits for it).
 *                   (Measured alternatives, round 6: a device-side snapshot + the D2H on a side stream under the next train is
 *                   SLOWER -- the D2H into registered host memory is a blit KERNEL that competes with the train's kernels:
 *                   0.29 vs 0.27 ms per 128-frame train --, and the same D2H enqueued later by a helper thread serialises with
 *                   the learner thread's next launch.)
 *   tail_in_graph   (ABI >= 12; honoured with wait_loss != 0 and loss_host set; wait_loss == 2: do not wait, see xt_net_io_wait)
 *                   1: the loss read-back and the parameter copy become the last two KERNELS OF THE TRAIN ITSELF -- inside the
 *                   replayed hipGraph, whose kernel arguments are fixed: the learner thread writes {destination, sequence
 *                   number} into a 64-byte page-locked mailbox the library owns; the first tail kernel reads the mailbox
 *                   over the bus, writes the loss sums + the sequence number back into it (and the sums to loss_acc) and hands
 *                   the destination to the copy kernel through device memory (so the host may rewrite the mailbox as soon as
 *                   it has seen the loss).  The chunks of such a train accumulate into 4 floats of the workspace that the loss
 *                   kernel re-arms: the graph holds no memset node.  The call returns when the sequence number has landed (a
 *                   bounded poll of the page-locked word, stream health checked every ~0.3 ms) and has copied the 4 floats to
 *                   loss_host.  consumed_event / publish_event are optional (an event record behind a replayed graph costs the
 *                   next graph ~10 us: the train's inputs are consumed once its loss has been seen, and the copy kernel reports
 *                   its own completion, xt_net_io_publish_wait); loss_event is not used.
 *                   2: the parameter copy leaves the GPU's critical path: the train's last kernel is a device-side SNAPSHOT
 *                   (params -> one of two buffers of the library, ~5 us, reported through the mailbox), and the 80 us bus-bound
 *                   copy snapshot -> publish_dst runs on the SDMA ENGINE under the next train, issued through the HSA runtime
 *                   by whoever calls xt_net_io_publish_wait for that train (a weights ring's committer thread; the learner
 *                   thread itself when it needs the buffer back, two publishes later).  publish_event must be NULL.  Why not a
 *                   copy kernel or hipMemcpyAsync on a side stream: csrc/xt_sdma.hip (both measured, round 6).  With plain Adam
 *                   and no gradient exchange both tail kernels are FOLDED into the Adam kernel of the train's last chunk (its
 *                   block 0 reports the loss before the update, every block writes the snapshot too, the last block reports it).
 *                   What 1 / 2 remove (rocprofv3 traces of the 128-frame loop): the 19-27 us between a replayed graph and the
 *                   next launch on the stream (twice), the wake-up of hipEventSynchronize, the host's share of the next
 *                   train's launch, and (2) the 74 us copy from between two trains: 210-235 -> ~194 (1) -> ~130 us of GPU time
 *                   per train + publish.
 * Events are hipEvent_t handles (e.g. torch.cuda.Event.cuda_event). */
typedef struct xt_train_io {
  void* wait_event;
  void* consumed_event;
  float* loss_host;
  void* loss_event;
  float* publish_dst;
  void* publish_event;
  int32_t wait_loss;
  int32_t tail_in_graph;
  uint64_t wait_dma_ticket;   /* != 0: the call waits (host side, before the launch) for xt_dma_wait_upto(ticket) */
} xt_train_io;
int xt_net_impala_train_io(xt_net* net, const xt_impala_cfg* cfg, const void* obs, int32_t n, int32_t batch_size,
                           const float* bp_logits, const int32_t* action, const uint8_t* done, const float* reward,
                           const float* lr_steps, float* loss_acc, int32_t use_graph, const xt_train_io* io, void* stream);
/* Second half of a train enqueued with io->tail_in_graph and io->wait_loss == 2 (ABI >= 12): xt_net_impala_train_io returned
 * right behind the launch -- the caller does whatever of its own book-keeping does not depend on the loss while the device
 * trains -- and this call waits (bounded poll of the mailbox, as above) for the loss of the MOST RECENT such train and copies
 * loss_acc's 4 floats to loss_host4 (any host memory). */
int xt_net_io_wait(xt_net* net, float* loss_host4, void* stream);

/* The parameter copy of a tail_in_graph train reports its own completion (ABI >= 12): the copy kernel's last workgroup writes
 * the train's sequence number into the mailbox, so a caller that passes NO publish_event / consumed_event (event records behind
 * a replayed graph delay the next graph on the stream by ~20 us) still learns when the block at publish_dst is complete:
 * xt_net_io_seq = sequence number of the most recent tail_in_graph train of this net (read it right behind the launch);
 * xt_net_io_publish_wait returns 0 once the copy of train `seq` (or a later one) has landed, 1 when timeout_ms ran out
 * (0 = query, < 0 = no limit).  For a tail_in_graph = 2 train it waits for the snapshot's report and then MAKES the copy (SDMA
 * engine, synchronous for the caller; the first thread that comes for a publish makes it, others wait for it; the query form
 * never starts it).  The train's inputs are consumed once its loss has been seen (the loss kernel runs behind every
 * kernel that reads them). */
uint32_t xt_net_io_seq(const xt_net* net);
/* 1 once the loss of the most recent tail_in_graph train has landed in the mailbox (xt_net_io_wait would return at once), else 0:
 * a learner thread that stages the next rollout message ITSELF while the device trains polls this between two messages */
int32_t xt_net_io_loss_ready(const xt_net* net);
int xt_net_io_publish_wait(xt_net* net, uint32_t seq, int32_t timeout_ms);

/* One synchronous device -> page-locked-host copy on the SDMA engine through the HSA runtime of the process (ABI >= 12;
 * csrc/xt_sdma.hip says why not hipMemcpyAsync).  dst_host: hipHostMalloc'ed or hipHostRegister'ed memory; src_dev: a device
 * allocation; the caller has made sure the source is complete (no stream is involved).  Diagnostic / test entry: the library
 * uses the same copy for the parameter block of xt_net_impala_train_io (tail_in_graph = 2). */
int xt_sdma_copy_d2h(void* dst_host, const void* src_dev, int64_t bytes);

/* Asynchronous page-locked-host -> device copies on the SDMA engines through the HSA runtime, with TICKETS instead of streams and
 * events (ABI >= 12): the rollout ingest's per-message H2D (hipMemcpyAsync + the event records around it cost the staging thread
 * ~30 us per message; this ~5).  xt_dma_h2d_async starts one copy and returns its ticket (tickets count up from 1; at most 64 in
 * flight); xt_dma_wait_upto returns 0 once EVERY copy with a ticket <= `ticket` has landed, 1 when timeout_ms ran out (0 = query,
 * < 0 = no limit).  No ordering against any HIP stream: the caller knows that the destination is free (xt_train_io: a train's
 * inputs are consumed once its loss has been seen) and waits for the ticket before it launches what reads the data
 * (xt_train_io.wait_dma_ticket). */
int xt_dma_h2d_async(void* dst_dev, const void* src_host, int64_t bytes, uint64_t* ticket_out);
int xt_dma_wait_upto(uint64_t ticket, int32_t timeout_ms);

/* diagnostic (ABI >= 12): host time (us, accumulated over *calls_out calls with a non-NULL io) of xt_net_impala_train_io's
 * phases -- [0] before the launch (wait for the copies, mailbox), [1] the launch (hipGraphLaunch, or the eager enqueue), [2] the
 * runtime calls behind it (event records, the separate copies when tail_in_graph is off), [3] the wait for the loss.
 * reset != 0 clears the accumulators. */
int xt_net_io_times(xt_net* net, double* us_out4, int64_t* calls_out, int32_t reset);

/* clip + Adam on the net's flat gradient (second half of a data-parallel step) */
int xt_net_apply(xt_net* net, float lr, float beta1, float beta2, float eps, float clip_norm,
                 float grad_scale, void* stream);

/* introspection for tests/tools: float offsets inside the bound workspace of layer `layer`'s post-activation
 * output [B,OH,OW,N] (out4[0]), its d(pre-activation) buffer (out4[1]), its weight-gradient slabs (out4[2]) and
 * their capacity in slabs (out4[3]) */
int xt_net_layer_offsets(const xt_net* net, int32_t layer, int64_t* out4);

/* arithmetic of the most recent layer launch made through this library by the calling process (diagnostic, for
 * bench.py's roofline: which matrix-core peak a timed kernel is priced against).  ABI >= 7. */
#define XT_ARITH_FP32 0          /* v_mfma_f32_32x32x2_f32                                                     */
#define XT_ARITH_BF16X3 1        /* uint8 x fp32 as 3 x v_mfma_f32_32x32x16_bf16 (first-layer kernels)          */
#define XT_ARITH_BF16X6 2        /* fp32 x fp32 as 6 x v_mfma_f32_32x32x16_bf16                                 */
#define XT_ARITH_FP32_BF16X6 3   /* fused backward launch: weight gradient fp32 MFMA, input gradient bf16x6     */
int32_t xt_last_launch_arith(void);

/* kernel-time probe: average duration (ms) of `reps` back-to-back launches of ONE layer kernel of the bound
 * network on `stream`, measured with HIP events on that stream (bench.py's roofline, tools/layer_bench.py). */
int xt_net_time_layer(xt_net* net, int32_t layer, int32_t which /* 0 fwd, 1 wgrad, 2 dgrad, 3 fused dgrad+wgrad */,
                      const void* obs, const int32_t* idx, int32_t B, int32_t reps,
                      float* ms_out, void* stream);

/* kernel-time probe of the SGD step's TAIL (ABI >= 11): average duration (ms, HIP events on `stream`) of `reps` back-to-back
 * repetitions of what follows the backward pass -- gradient reduction, [data parallel: exchange], global-norm clip + Adam --
 * in the net's CURRENT mode (plain; exchange hook; xt_net_set_dp + xt_net_set_direct: the fused three-launch chain), on the
 * weight-gradient slabs the most recent gradient step left in the workspace (parameters and slots ARE updated).  N processes
 * running it at the same time on one device give the kernel-side cost of the exchange per rank (tools/direct_probe.py). */
int xt_net_time_tail(xt_net* net, float lr, float clip_norm, int32_t reps, float* ms_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XT_MI355X_H_ */
