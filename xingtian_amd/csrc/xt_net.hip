// Network-level driver: one C call enqueues a whole Model.train() (all epochs and
// minibatches of xt/model/ppo/ppo.py:111-132, or one ImpalaCnnOpt.train chunk) on a HIP
// stream, optionally captured once into a hipGraph and replayed (a B=320 step is ~20
// short kernels; the reference pays a feed_dict H2D + session dispatch per minibatch).
#include <vector>
#include <string>
#include <string.h>

#include "xt_common.h"

namespace xt {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int launch_fwd(const xt_conv_geom*, const xt_input_xform*, int, const void*, const int32_t*, const float*,
               const float*, float*, float*, int, hipStream_t);
int launch_wgrad(const xt_conv_geom*, const xt_input_xform*, int, const void*, const int32_t*, const float*,
                 float*, float*, int, hipStream_t);
int launch_dgrad(const xt_conv_geom*, int, const float*, const float*, const float*, int, float*, hipStream_t);
int launch_global_norm(const float*, long long, float, float, float, float, float, int, float*, float*, hipStream_t);
int launch_adam(float*, const float*, float*, float*, long long, float, float, float, const float*, hipStream_t);

struct Layer {
  xt_conv_geom g;
  int64_t poff;
  int trunk;
  int K, OHOW;
  int64_t act_off, dact_off;   // floats into the workspace
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
  int64_t off_partial, off_slabs, off_logits, off_value, off_dlogits, off_dvalue, off_terms, off_loss, off_norm;
  int64_t partial_floats, slab_floats;
  // graph cache for ppo_train
  hipGraphExec_t gexec = nullptr;
  hipStream_t cap_stream = nullptr;   // capture happens on our own stream: the legacy null stream cannot be captured
  std::string gkey;
};

namespace xt {

static int fwd_split(const Layer& L, int B) {
  const int M = B * L.OHOW, N = L.g.N;
  const int tiles = (N <= 32) ? ((M + 127) / 128) * ((N + 31) / 32) : ((M + 63) / 64) * ((N + 63) / 64);
  const int ksteps = (L.K + 31) / 32;
  if (tiles >= 128 || ksteps < 4) return 1;
  int s = 512 / tiles;
  if (s > ksteps / 2) s = ksteps / 2;
  return s < 1 ? 1 : s;
}
static int wgrad_split(const Layer& L, int B) {
  const int M = B * L.OHOW, N = L.g.N;
  const int tiles = (N <= 32) ? ((L.K + 127) / 128) * ((N + 31) / 32) : ((L.K + 63) / 64) * ((N + 63) / 64);
  const int msteps = (M + 31) / 32;
  int s = 512 / tiles;
  if (s > msteps / 4) s = msteps / 4;
  return s < 1 ? 1 : s;
}

static int net_forward(xt_net* n, const void* obs, const int32_t* idx, int B, hipStream_t st) {
  for (int tr = 0; tr < n->n_trunks; ++tr) {
    const void* x = obs;
    for (int l = n->t_begin[tr]; l < n->t_end[tr]; ++l) {
      Layer& L = n->layers[l];
      const bool first = (l == n->t_begin[tr]);
      if (int rc = launch_fwd(&L.g, first ? &n->xf : nullptr, B, x, first ? idx : nullptr, n->params + L.poff,
                              n->params + L.poff + (int64_t)L.K * L.g.N, n->ws + L.act_off,
                              n->ws + n->off_partial, fwd_split(L, B), st))
        return rc;
      x = n->ws + L.act_off;
    }
  }
  const float* f_pi = n->ws + n->layers[n->t_end[0] - 1].act_off;
  const float* f_v = n->ws + n->layers[n->t_end[n->n_trunks - 1] - 1].act_off;
  const int F = n->feat, A = n->A;
  return xt_heads_fwd(f_pi, f_v, B, F, A, n->params + n->pi_off, n->params + n->pi_off + (int64_t)F * A,
                      n->params + n->v_off, n->params + n->v_off + F, n->ws + n->off_logits, n->ws + n->off_value, st);
}

// heads backward + trunk backward (dlogits/dvalue already in the workspace)
static int net_backward(xt_net* n, const void* obs, const int32_t* idx, int B, hipStream_t st) {
  const int F = n->feat, A = n->A;
  Layer& Lp = n->layers[n->t_end[0] - 1];
  Layer& Lv = n->layers[n->t_end[n->n_trunks - 1] - 1];
  if (int rc = xt_heads_bwd(n->ws + Lp.act_off, n->ws + Lv.act_off, B, F, A, n->params + n->pi_off,
                            n->params + n->v_off, n->ws + n->off_dlogits, n->ws + n->off_dvalue, Lp.g.act,
                            n->grads + n->pi_off, n->grads + n->pi_off + (int64_t)F * A, n->grads + n->v_off,
                            n->grads + n->v_off + F, n->ws + Lp.dact_off, n->ws + Lv.dact_off, st))
    return rc;
  for (int tr = 0; tr < n->n_trunks; ++tr) {
    for (int l = n->t_end[tr] - 1; l >= n->t_begin[tr]; --l) {
      Layer& L = n->layers[l];
      const bool first = (l == n->t_begin[tr]);
      const void* x = first ? obs : (const void*)(n->ws + n->layers[l - 1].act_off);
      if (int rc = launch_wgrad(&L.g, first ? &n->xf : nullptr, B, x, first ? idx : nullptr, n->ws + L.dact_off,
                                n->grads + L.poff, n->ws + n->off_slabs, wgrad_split(L, B), st))
        return rc;
      if (!first) {
        Layer& Lprev = n->layers[l - 1];
        if (int rc = launch_dgrad(&L.g, B, n->ws + L.dact_off, n->params + L.poff, n->ws + Lprev.act_off,
                                  Lprev.g.act, n->ws + Lprev.dact_off, st))
          return rc;
      }
    }
  }
  return 0;
}

static int net_apply(xt_net* n, float lr, float b1, float b2, float eps, float clip, float gscale, hipStream_t st) {
  if (int rc = launch_global_norm(n->grads, n->P, clip, gscale, lr, b1, b2, 1, n->state, n->ws + n->off_norm, st))
    return rc;
  return launch_adam(n->params, n->grads, n->m, n->v, n->P, b1, b2, eps, n->state, st);
}

static int ppo_step(xt_net* n, const xt_ppo_cfg* c, const void* obs, const int32_t* idx, int B,
                    const int32_t* action, const float* old_logp, const double* adv, const float* old_v,
                    const double* target_v, int apply, float* loss_out, float* loss_acc, hipStream_t st) {
  XT_REQUIRE(n->params && n->ws, "xt_net: buffers not bound (call xt_net_bind)");
  XT_REQUIRE(B > 0 && B <= n->maxB, "xt_net_ppo_step: batch %d outside (0,%d]", B, n->maxB);
  if (int rc = net_forward(n, obs, idx, B, st)) return rc;
  const float inv_b = 1.f / (float)(c->global_batch > 0 ? c->global_batch : B);
  if (int rc = xt_ppo_loss(n->ws + n->off_logits, n->ws + n->off_value, B, n->A, idx, action, old_logp, adv, old_v,
                           target_v, c->clip_ratio, c->ent_coef, c->vf_clip, c->critic_coef, inv_b,
                           n->ws + n->off_dlogits, n->ws + n->off_dvalue, n->ws + n->off_terms, st))
    return rc;
  float* lo = loss_out ? loss_out : n->ws + n->off_loss;
  if (int rc = xt_ppo_loss_reduce(n->ws + n->off_terms, B, c->ent_coef, c->critic_coef, inv_b, lo, loss_acc, st))
    return rc;
  if (int rc = net_backward(n, obs, idx, B, st)) return rc;
  if (apply)
    return net_apply(n, c->lr, c->beta1, c->beta2, c->eps, c->max_grad_norm, c->grad_scale, st);
  return 0;
}

}  // namespace xt

extern "C" {

int xt_abi_version(void) { return XT_ABI_VERSION; }
const char* xt_last_error(void) { return xt::g_err; }
const char* xt_build_arch(void) { return "gfx950"; }

int xt_net_create(const xt_net_desc* d, int32_t max_batch, xt_net** out) {
  XT_REQUIRE(d && out && max_batch > 0, "xt_net_create: bad arguments");
  XT_REQUIRE(d->n_trunks == 1 || d->n_trunks == 2, "xt_net_create: n_trunks must be 1 or 2");
  XT_REQUIRE(d->n_layers >= d->n_trunks, "xt_net_create: every trunk needs at least one layer");
  xt_net* n = new xt_net();
  n->n_trunks = d->n_trunks; n->feat = d->feat; n->A = d->action_dim;
  n->pi_off = d->pi_off; n->v_off = d->v_off; n->P = d->n_params; n->xf = d->xf;
  n->in_h = d->in_h; n->in_w = d->in_w; n->in_c = d->in_c; n->maxB = max_batch;
  int64_t off = 0;
  int64_t max_partial = 4, max_slab = 4;
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
    n->layers.push_back(L);
    // wgrad slab bound: msplit <= max(1, 512/tiles) slabs of (K+1)*N floats
    const int tiles = (L.g.N <= 32) ? ((L.K + 127) / 128) : ((L.K + 63) / 64) * ((L.g.N + 63) / 64);
    const int64_t s = (int64_t)((512 / tiles) < 1 ? 1 : (512 / tiles)) * (int64_t)(L.K + 1) * L.g.N;
    if (s > max_slab) max_slab = s;
  }
  if (d->n_trunks == 2 && n->t_begin[1] == 0) {
    delete n;
    XT_REQUIRE(false, "xt_net_create: trunk 1 has no layers");
  }
  // fwd split-K partial bound: ksplit*tiles <= 512 and a tile holds 4096 outputs
  max_partial = (int64_t)512 * 4096;
  n->partial_floats = xt::align4(max_partial); n->slab_floats = xt::align4(max_slab);
  n->off_partial = off; off += n->partial_floats;
  n->off_slabs = off; off += n->slab_floats;
  n->off_logits = off; off += xt::align4((int64_t)max_batch * n->A);
  n->off_value = off; off += xt::align4(max_batch);
  n->off_dlogits = off; off += xt::align4((int64_t)max_batch * n->A);
  n->off_dvalue = off; off += xt::align4(max_batch);
  n->off_terms = off; off += xt::align4((int64_t)max_batch * 4);
  n->off_loss = off; off += xt::align4(8 + max_batch);
  n->off_norm = off; off += 1024;
  n->ws_floats = off;
  *out = n;
  return 0;
}

void xt_net_destroy(xt_net* net) {
  if (!net) return;
  if (net->gexec) hipGraphExecDestroy(net->gexec);
  if (net->cap_stream) hipStreamDestroy(net->cap_stream);
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
  n->params = params; n->grads = grads; n->m = adam_m; n->v = adam_v; n->state = adam_state;
  n->ws = static_cast<float*>(workspace);
  if (n->gexec) { hipGraphExecDestroy(n->gexec); n->gexec = nullptr; n->gkey.clear(); }
  return 0;
}

int xt_net_forward(xt_net* n, const void* obs, const int32_t* idx, int32_t B, float* logits, float* value,
                   void* stream) {
  XT_REQUIRE(n && n->params && n->ws, "xt_net_forward: buffers not bound");
  XT_REQUIRE(B > 0 && B <= n->maxB, "xt_net_forward: batch %d outside (0,%d]", B, n->maxB);
  hipStream_t st = xt::as_stream(stream);
  if (int rc = xt::net_forward(n, obs, idx, B, st)) return rc;
  if (logits) XT_CHECK_HIP(hipMemcpyAsync(logits, n->ws + n->off_logits, sizeof(float) * B * n->A, hipMemcpyDeviceToDevice, st));
  if (value) XT_CHECK_HIP(hipMemcpyAsync(value, n->ws + n->off_value, sizeof(float) * B, hipMemcpyDeviceToDevice, st));
  return 0;
}

int xt_net_ppo_step(xt_net* net, const xt_ppo_cfg* cfg, const void* obs, const int32_t* idx, int32_t B,
                    const int32_t* action, const float* old_logp, const double* adv, const float* old_v,
                    const double* target_v, int32_t apply, float* loss_out, float* loss_acc, void* stream) {
  XT_REQUIRE(net && cfg, "xt_net_ppo_step: null argument");
  return xt::ppo_step(net, cfg, obs, idx, B, action, old_logp, adv, old_v, target_v, apply, loss_out, loss_acc,
                      xt::as_stream(stream));
}

static int ppo_train_enqueue(xt_net* net, const xt_ppo_cfg* c, const void* obs, int32_t n, const int32_t* perm,
                             const int32_t* action, const float* old_logp, const double* adv, const float* old_v,
                             const double* target_v, float* loss_acc, hipStream_t st) {
  XT_CHECK_HIP(hipMemsetAsync(loss_acc, 0, 2 * sizeof(float), st));
  for (int ep = 0; ep < c->num_sgd_iter; ++ep) {
    for (int start = 0; start < n; start += c->batch_size) {
      const int B = (n - start) < c->batch_size ? (n - start) : c->batch_size;
      xt_ppo_cfg cc = *c;
      if (cc.global_batch > 0 && B != c->batch_size)   // short last minibatch: keep the local/global ratio
        cc.global_batch = (int)((long long)cc.global_batch * B / c->batch_size);
      if (int rc = xt::ppo_step(net, &cc, obs, perm + (size_t)ep * n + start, B, action, old_logp, adv, old_v,
                                target_v, 1, nullptr, loss_acc, st))
        return rc;
    }
  }
  return 0;
}

int xt_net_ppo_train(xt_net* net, const xt_ppo_cfg* c, const void* obs, int32_t n, const int32_t* perm,
                     const int32_t* action, const float* old_logp, const double* adv, const float* old_v,
                     const double* target_v, float* loss_acc, int32_t use_graph, void* stream) {
  XT_REQUIRE(net && c && obs && perm && loss_acc, "xt_net_ppo_train: null argument");
  XT_REQUIRE(n > 0 && c->batch_size > 0 && c->batch_size <= net->maxB && c->num_sgd_iter > 0,
             "xt_net_ppo_train: bad sizes (n=%d batch=%d max=%d)", n, c->batch_size, net->maxB);
  hipStream_t st = xt::as_stream(stream);
  if (!use_graph)
    return ppo_train_enqueue(net, c, obs, n, perm, action, old_logp, adv, old_v, target_v, loss_acc, st);
  char key[512];
  snprintf(key, sizeof(key), "%p|%d|%p|%p|%p|%p|%p|%p|%p|%g|%g|%g|%g|%g|%g|%g|%g|%g|%d|%d|%g|%d", obs, n,
           (const void*)perm, (const void*)action, (const void*)old_logp, (const void*)adv, (const void*)old_v,
           (const void*)target_v, (void*)loss_acc, c->lr, c->beta1, c->beta2, c->eps, c->clip_ratio, c->ent_coef,
           c->vf_clip, c->critic_coef, c->max_grad_norm, c->batch_size, c->num_sgd_iter, c->grad_scale,
           c->global_batch);
  if (!net->gexec || net->gkey != key) {
    if (net->gexec) { hipGraphExecDestroy(net->gexec); net->gexec = nullptr; }
    hipGraph_t graph = nullptr;
    if (!net->cap_stream) XT_CHECK_HIP(hipStreamCreateWithFlags(&net->cap_stream, hipStreamNonBlocking));
    hipStream_t cs = net->cap_stream;
    XT_CHECK_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
    int rc = ppo_train_enqueue(net, c, obs, n, perm, action, old_logp, adv, old_v, target_v, loss_acc, cs);
    hipError_t e = hipStreamEndCapture(cs, &graph);
    if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
    XT_CHECK_HIP(e);
    e = hipGraphInstantiate(&net->gexec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    XT_CHECK_HIP(e);
    net->gkey = key;
  }
  XT_CHECK_HIP(hipGraphLaunch(net->gexec, st));
  return 0;
}

int xt_net_impala_step(xt_net* n, const xt_impala_cfg* c, const void* obs, int32_t nfr, const float* bp_logits,
                       const int32_t* action, const uint8_t* done, const float* reward, int32_t apply,
                       float* loss_out, float* loss_acc, void* stream) {
  XT_REQUIRE(n && c && n->params && n->ws, "xt_net_impala_step: null argument / unbound buffers");
  const int T = c->sample_batch_step;
  XT_REQUIRE(T >= 2 && nfr > 0 && nfr % T == 0, "xt_net_impala_step: n=%d must be a multiple of sample_batch_step=%d",
             nfr, T);
  XT_REQUIRE(nfr <= n->maxB, "xt_net_impala_step: %d frames > max batch %d", nfr, n->maxB);
  hipStream_t st = xt::as_stream(stream);
  if (int rc = xt::net_forward(n, obs, nullptr, nfr, st)) return rc;
  float* lo = n->ws + n->off_loss;   // needs 4 + n_traj floats
  if (int rc = xt_impala_loss(n->ws + n->off_logits, n->ws + n->off_value, bp_logits, action, done, reward, nfr / T,
                              T, n->A, c->gamma, n->ws + n->off_dlogits, n->ws + n->off_dvalue, lo, loss_acc,
                              nullptr, nullptr, st))
    return rc;
  if (loss_out) XT_CHECK_HIP(hipMemcpyAsync(loss_out, lo, sizeof(float), hipMemcpyDeviceToDevice, st));
  if (int rc = xt::net_backward(n, obs, nullptr, nfr, st)) return rc;
  if (apply) return xt::net_apply(n, c->lr, c->beta1, c->beta2, c->eps, c->grad_norm_clip, c->grad_scale, st);
  return 0;
}

int xt_net_apply(xt_net* n, float lr, float beta1, float beta2, float eps, float clip_norm, float grad_scale,
                 void* stream) {
  XT_REQUIRE(n && n->params && n->m && n->v && n->state, "xt_net_apply: buffers not bound");
  return xt::net_apply(n, lr, beta1, beta2, eps, clip_norm, grad_scale, xt::as_stream(stream));
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
  XT_REQUIRE(which == 0 || which == 1 || (which == 2 && !first), "xt_net_time_layer: bad kernel selector");
  hipEvent_t e0, e1;
  XT_CHECK_HIP(hipEventCreate(&e0));
  XT_CHECK_HIP(hipEventCreate(&e1));
  int rc = 0;
  auto one = [&]() -> int {
    if (which == 0)
      return xt::launch_fwd(&L.g, first ? &n->xf : nullptr, B, x, first ? idx : nullptr, n->params + L.poff,
                            n->params + L.poff + (int64_t)L.K * L.g.N, n->ws + L.act_off, n->ws + n->off_partial,
                            xt::fwd_split(L, B), st);
    if (which == 1)
      return xt::launch_wgrad(&L.g, first ? &n->xf : nullptr, B, x, first ? idx : nullptr, n->ws + L.dact_off,
                              n->grads + L.poff, n->ws + n->off_slabs, xt::wgrad_split(L, B), st);
    xt::Layer& Lp = n->layers[layer - 1];
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
