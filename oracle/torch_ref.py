"""Oracle (test infrastructure): torch-CPU autograd restatement of the same update.

Two uses, both as a *checker / baseline*, never as product code:

1. ``tests/test_oracle.py`` cross-checks the hand-derived numpy gradients of
   ``oracle/nets.py`` against torch autograd in float64.
2. ``bench.py``'s ``cpu_baseline`` leg times this float32 path (oneDNN, all host
   cores) as the stand-in for the reference's TF-CPU learner, which cannot run
   here (TensorFlow is not installed): same loop structure as
   xt/model/ppo/ppo.py:111-132 incl. the per-minibatch fancy-index gather.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _to_t(a, dtype):
    return torch.as_tensor(np.asarray(a), dtype=dtype)


class TorchActorCritic(object):
    def __init__(self, spec, params, dtype=torch.float64):
        self.spec, self.dtype = spec, dtype
        self.params = {k: _to_t(v, dtype).clone().requires_grad_(True) for k, v in params.items()}

    def _transform(self, obs):
        sc = self.spec["input_scale"]
        x = torch.as_tensor(obs)
        if sc is None or x.dtype != torch.uint8:
            return x.to(self.dtype)
        x = x.to(self.dtype)
        kind, mean, std = sc
        if kind == "div" or abs(mean) < 1e-4:
            return x / std
        return (x - mean) / std

    def forward(self, obs):
        x0 = self._transform(obs)
        b = x0.shape[0]
        feats = []
        for trunk in self.spec["trunks"]:
            x = x0
            for lay in trunk:
                w = self.params[lay.name + "/kernel"]
                bias = self.params[lay.name + "/bias"]
                if lay.kind == "conv":
                    xin = x.reshape(b, lay.in_h, lay.in_w, lay.cin).permute(0, 3, 1, 2)
                    xin = F.pad(xin, (lay.pl, lay.pr, lay.pt, lay.pb))
                    y = F.conv2d(xin, w.permute(3, 2, 0, 1), bias, stride=lay.s)
                    y = y.permute(0, 2, 3, 1)  # back to NHWC (Flatten order H,W,C)
                else:
                    y = x.reshape(b, -1) @ w + bias
                x = _ACT[lay.act](y)
            feats.append(x.reshape(b, -1))
        f_pi, f_v = feats[0], feats[-1]
        wpi = self.params[self.spec["pi_name"] + "/kernel"].reshape(f_pi.shape[1], -1)
        logits = f_pi @ wpi + self.params[self.spec["pi_name"] + "/bias"]
        value = f_v @ self.params[self.spec["v_name"] + "/kernel"] + self.params[self.spec["v_name"] + "/bias"]
        return logits, value


_ACT = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "softsign": F.softsign, "softplus": lambda y: torch.logaddexp(y, torch.zeros_like(y)),   # (F.softplus switches to the identity above 20)
        
        "leaky_relu": lambda y: F.leaky_relu(y, 0.2), "elu": F.elu, "selu": F.selu, "swish": F.silu,
        "gelu": lambda y: F.gelu(y, approximate="tanh"), None: lambda y: y, "none": lambda y: y}


def ppo_loss_torch(logits, value, action, old_logp, adv, old_v, target_v,
                   clip_ratio, ent_coef, vf_clip, critic_coef):
    """xt/model/ppo/__init__.py:4-25 + tf_dist.py:103-113 in torch ops."""
    a = torch.as_tensor(np.asarray(action), dtype=torch.int64).reshape(-1)
    logp = -F.cross_entropy(logits, a, reduction="none").unsqueeze(-1)
    ratio = torch.exp(logp - old_logp)
    surr = torch.minimum(ratio * adv, torch.clamp(ratio, 1.0 - clip_ratio, 1.0 + clip_ratio) * adv)
    rl = logits - logits.max(dim=-1, keepdim=True).values
    e = torch.exp(rl)
    z = e.sum(dim=-1, keepdim=True)
    ent = ((e / z) * (torch.log(z) - rl)).sum(dim=-1, keepdim=True)
    actor = -surr.mean() - ent_coef * ent.mean()
    vf1 = (value - target_v) ** 2
    vclip = old_v + torch.clamp(value - old_v, -vf_clip, vf_clip)
    vf2 = (vclip - target_v) ** 2
    critic = 0.5 * torch.maximum(vf1, vf2).mean()
    return actor + critic_coef * critic


def impala_loss_torch(logits, baseline, bp_logits, actions, dones, rewards, batch_step, gamma=0.99):
    """impala_cnn_opt.py:188-196,299-351 + vtrace.py:39-115 in torch ops."""
    dt = logits.dtype

    def split(t, drop_last=False):
        bc = t.shape[0] // batch_step
        r = t.reshape((bc, batch_step) + tuple(t.shape[1:])).transpose(0, 1)
        return r[:-1] if drop_last else r

    tp = split(logits, True)
    bp = split(torch.as_tensor(np.asarray(bp_logits), dtype=dt), True)
    act = split(torch.as_tensor(np.asarray(actions), dtype=torch.int64), True)
    disc = split((~torch.as_tensor(np.asarray(dones, bool))).to(dt) * gamma, True)
    rew = split(torch.clamp(torch.as_tensor(np.asarray(rewards), dtype=dt), -1, 1), True)
    vals = split(baseline, True)
    boot = split(baseline)[-1]
    a_dim = tp.shape[-1]
    ce_t = F.cross_entropy(tp.reshape(-1, a_dim), act.reshape(-1), reduction="none").reshape(act.shape)
    ce_b = F.cross_entropy(bp.reshape(-1, a_dim), act.reshape(-1), reduction="none").reshape(act.shape)
    with torch.no_grad():
        rhos = torch.exp(-ce_t + ce_b)
        crho = torch.clamp(rhos, max=1.0)
        cs = torch.clamp(rhos, max=1.0)
        nv = torch.cat([vals[1:], boot[None]], 0)
        deltas = crho * (rew + disc * nv - vals)
        acc = torch.zeros_like(boot)
        outs = []
        for t in range(vals.shape[0] - 1, -1, -1):
            acc = deltas[t] + disc[t] * cs[t] * acc
            outs.append(acc)
        vs = torch.stack(outs[::-1], 0) + vals
        vsn = torch.cat([vs[1:], boot[None]], 0)
        pg_adv = crho * (rew + disc * vsn - vals)
    pi_loss = (ce_t * pg_adv).sum()
    val_loss = 0.5 * ((vs - vals) ** 2).sum()
    ent_loss = -(-(F.softmax(tp, -1) * F.log_softmax(tp, -1)).sum(-1)).sum()
    return pi_loss + 0.5 * val_loss + 0.01 * ent_loss


class TorchAdamTF(object):
    """TF1 Adam on a list of tensors (in place, no autograd)."""

    def __init__(self, params, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.params, self.lr, self.b1, self.b2, self.eps, self.t = params, lr, b1, b2, eps, 0
        self.m = {k: torch.zeros_like(p) for k, p in params.items()}
        self.v = {k: torch.zeros_like(p) for k, p in params.items()}

    @torch.no_grad()
    def apply(self, grads, clip_norm):
        self.t += 1
        gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
        scale = clip_norm / max(gn, clip_norm)
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k, p in self.params.items():
            g = grads[k] * scale
            self.m[k] += (g - self.m[k]) * (1.0 - self.b1)
            self.v[k] += (g * g - self.v[k]) * (1.0 - self.b2)
            p -= lr_t * self.m[k] / (torch.sqrt(self.v[k]) + self.eps)
        return gn


def gauss_ppo_loss_torch(mean, log_std, value, action, old_logp, adv, old_v, target_v,
                         clip_ratio, ent_coef, vf_clip, critic_coef):
    """DiagGaussianDist (tf_dist.py:47-87) in the PPO loss, written with torch ops for autograd."""
    x = torch.as_tensor(np.asarray(action), dtype=mean.dtype).reshape(mean.shape)
    param_ls = mean * 0.0 + log_std                      # ppo.py:79
    std = torch.exp(param_ls)
    neglogp = 0.5 * np.log(2.0 * np.pi) * mean.shape[-1] + 0.5 * (((x - mean) / std) ** 2).sum(-1, keepdim=True) \
        + param_ls.sum(-1, keepdim=True)
    logp = -neglogp
    ent = (param_ls + 0.5 * (np.log(2.0 * np.pi) + 1.0)).sum(-1, keepdim=True)
    ratio = torch.exp(logp - old_logp)
    surr = torch.minimum(ratio * adv, torch.clamp(ratio, 1.0 - clip_ratio, 1.0 + clip_ratio) * adv)
    actor = -surr.mean() - ent_coef * ent.mean()
    vf1 = (value - target_v) ** 2
    vclip = old_v + torch.clamp(value - old_v, -vf_clip, vf_clip)
    vf2 = (vclip - target_v) ** 2
    critic = 0.5 * torch.maximum(vf1, vf2).mean()
    return actor + critic_coef * critic


class TorchPpoLearner(object):
    """Whole PPO update on CPU; ``train`` mirrors xt/model/ppo/ppo.py:111-132."""

    def __init__(self, spec, params, cfg, dtype=torch.float32):
        self.net = TorchActorCritic(spec, params, dtype)
        self.cfg, self.dtype = cfg, dtype
        self.opt = TorchAdamTF(self.net.params, cfg["LR"])

    def step(self, obs, action, old_logp, adv, old_v, target_v, apply=True):
        c, dt = self.cfg, self.dtype
        for p in self.net.params.values():
            p.grad = None
        logits, value = self.net.forward(obs)
        if self.net.spec.get("action_type") == "DiagGaussian":
            loss = gauss_ppo_loss_torch(logits, self.net.params["pi_logstd"], value, action, _to_t(old_logp, dt),
                                        _to_t(adv, dt), _to_t(old_v, dt), _to_t(target_v, dt), c["LOSS_CLIPPING"],
                                        c["ENTROPY_LOSS"], c["VF_CLIP"], c["CRITIC_LOSS_COEF"])
        else:
            loss = ppo_loss_torch(logits, value, action, _to_t(old_logp, dt), _to_t(adv, dt), _to_t(old_v, dt),
                                  _to_t(target_v, dt), c["LOSS_CLIPPING"], c["ENTROPY_LOSS"], c["VF_CLIP"],
                                  c["CRITIC_LOSS_COEF"])
        loss.backward()
        grads = {k: p.grad.clone() for k, p in self.net.params.items()}
        gn = None
        if apply:
            gn = self.opt.apply(grads, c["MAX_GRAD_NORM"])
        return loss.item(), grads, gn

    def train(self, state, label, perms):
        obs = state[0]
        nbatch, bs = obs.shape[0], self.cfg["BATCH_SIZE"]
        losses = []
        for ep in range(self.cfg["NUM_SGD_ITER"]):
            inds = np.asarray(perms[ep])
            for start in range(0, nbatch, bs):
                mb = inds[start:start + bs]
                loss, _, _ = self.step(obs[mb], label[0][mb], label[1][mb], label[2][mb], label[3][mb],
                                       label[4][mb])
                losses.append(loss)
        return float(np.mean(losses))


class TorchImpalaLearner(object):
    def __init__(self, spec, params, cfg, dtype=torch.float32):
        self.net = TorchActorCritic(spec, params, dtype)
        self.cfg, self.dtype = cfg, dtype
        self.opt = TorchAdamTF(self.net.params, cfg["LR"])

    def step(self, state, bp_logits, actions, dones, rewards, apply=True):
        c = self.cfg
        for p in self.net.params.values():
            p.grad = None
        logits, value = self.net.forward(state)
        loss = impala_loss_torch(logits, value[:, 0], bp_logits, actions, dones, rewards,
                                 c["sample_batch_step"], c.get("GAMMA", 0.99))
        loss.backward()
        grads = {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
                 for k, p in self.net.params.items()}
        gn = None
        if apply:
            gn = self.opt.apply(grads, c["grad_norm_clip"])
        return loss.item(), grads, gn


def keras_impala_loss_torch(logits, value, adv, onehot, target_v, ent_coef):
    """impala_cnn.py:94-108 + compile(loss_weights) in torch ops (independent of oracle/nets.py):
    mean over [B,A] of adv*(-y*log(p+1e-10)) - ENT*(-p*log(p+1e-10)), plus 0.5 * mse(value, target)."""
    p = torch.softmax(logits, dim=-1)
    logp = torch.log(p + 1e-10)
    pi = (adv.reshape(-1, 1) * (-onehot * logp) - ent_coef * (-p * logp)).mean()
    mse = ((value.reshape(-1) - target_v.reshape(-1)) ** 2).mean()
    return pi + 0.5 * mse


class TorchKerasImpalaLearner(object):
    """Keras-form IMPALA update with torch autograd; optimizer = tf.keras Adam written out (per-tensor clip_by_norm,
    time-decayed lr, eps 1e-7 outside the square root)."""

    def __init__(self, spec, params, lr, ent_coef, clipnorm=None, decay=0.0, dtype=torch.float64):
        self.net = TorchActorCritic(spec, params, dtype)
        self.lr, self.ent, self.clipnorm, self.decay, self.dtype = lr, ent_coef, clipnorm, decay, dtype
        self.it = 0
        self.m = {k: torch.zeros_like(p) for k, p in self.net.params.items()}
        self.v = {k: torch.zeros_like(p) for k, p in self.net.params.items()}

    def step(self, obs, adv, onehot, target_v):
        dt = self.dtype
        for p in self.net.params.values():
            p.grad = None
        logits, value = self.net.forward(obs)
        loss = keras_impala_loss_torch(logits, value, _to_t(adv, dt), _to_t(onehot, dt), _to_t(target_v, dt), self.ent)
        loss.backward()
        grads = {k: p.grad.clone() for k, p in self.net.params.items()}
        lr = self.lr / (1.0 + self.decay * self.it)
        t = self.it + 1
        lr_t = lr * (1.0 - 0.999 ** t) ** 0.5 / (1.0 - 0.9 ** t)
        with torch.no_grad():
            for k, p in self.net.params.items():
                g = grads[k]
                if self.clipnorm is not None:
                    n = torch.linalg.vector_norm(g)
                    if n > self.clipnorm:
                        g = g * (self.clipnorm / n)
                self.m[k] = 0.9 * self.m[k] + 0.1 * g
                self.v[k] = 0.999 * self.v[k] + 0.001 * g * g
                p -= lr_t * self.m[k] / (torch.sqrt(self.v[k]) + 1e-7)
        self.it += 1
        return loss.item(), grads
