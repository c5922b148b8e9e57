"""Oracle (test infrastructure): golden vectors for the TensorFlow-side formulas, produced by
EXECUTING the reference's own sources under the torch-float64 ``tf`` stand-in of ``oracle/tf_shim.py``.

Runs only in the authoring container (needs /root/reference, read-only).  Executed unmodified:

* ``xt/model/ppo/__init__.py``      (actor_loss_with_entropy :4-17, critic_loss :20-26)
* ``xt/model/tf_dist.py``           (DiagGaussianDist :49-86, CategoricalDist :89-130, make_dist)
* ``xt/model/impala/vtrace.py``     (from_logic_outputs :39-115)
* ``xt/model/impala/impala_cnn_opt.py``: the module-level loss functions (calc_baseline_loss,
  calc_entropy_loss, calc_pi_loss, vtrace_loss :299-351) by importing the module under stubs, and the
  ``split_batches`` closure together with the ``self.loss = vtrace_loss(...)`` wiring (:166-196) by
  exec-ing exactly those source lines of ``create_model`` (text taken from the file, dedented, nothing
  edited) with a stand-in ``self`` that carries the placeholders.

The composition ``loss = actor_loss + CRITIC_LOSS_COEF * critic_loss`` (xt/model/ppo/ppo.py:89-92) and the
DiagGaussian ``dist_param = concat([pi_latent, pi_latent * 0.0 + log_std])`` (:75-79) are three lines
restated here (they live inside ``build_graph`` next to session/placeholder code).

* ``xt/model/impala/impala_cnn.py`` :99-108 and ``xt/model/impala/impala_mlp.py`` :84-93: the Keras-form
  ``impala_loss(advantage)`` closures, by exec-ing exactly those source lines with ``K`` = a three-function
  Keras-backend stand-in over torch float64 (``KerasBackend`` below) and the reference's own ``ENTROPY_LOSS``.
  What Keras does AROUND the closure -- the batch mean of the per-sample loss vector, ``'mse'`` for the value
  head, ``loss_weights`` [1.0, 0.5] (impala_cnn.py:59-62) -- is library behaviour restated in three lines.

Outputs (values and torch-autograd gradients, float64): ``tests/golden/tf_*.npz``.
Usage:  python oracle/gen_golden_tf.py
"""
import importlib.util
import os
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import tf_shim  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


class _Permissive(types.ModuleType):
    """A stub module: any name that is not set resolves to an inert placeholder."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()


class _Inert(object):
    def __call__(self, *a, **k):
        if len(a) == 1 and isinstance(a[0], type) and not k:
            return a[0]            # used as a class decorator (Registers.model)
        return _Inert()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference(tf):
    """Import the four reference files with ``tf`` = the shim; everything else they import is a stub."""
    names = ["xt", "xt.model", "xt.model.tf_compat", "xt.model.impala", "xt.model.impala.default_config",
             "xt.model.atari_model", "xt.model.tf_utils", "xt.model.model_utils", "xt.model.ppo",
             "zeus", "zeus.common", "zeus.common.util", "zeus.common.util.register", "zeus.common.util.common",
             "tensorflow", "tensorflow.python", "tensorflow.python.util", "absl", "absl.logging"]
    saved = {k: sys.modules.get(k) for k in names + ["xt.model.impala.vtrace", "xt.model.tf_dist"]}
    mods = {k: _Permissive(k) for k in names}
    for k in ("xt", "xt.model", "xt.model.impala", "zeus", "zeus.common", "zeus.common.util", "tensorflow",
              "tensorflow.python", "absl"):
        mods[k].__path__ = []
    for k in names:                               # "import a.b.c as d" walks attributes from the top package
        if "." in k:
            setattr(mods[k.rsplit(".", 1)[0]], k.rsplit(".", 1)[1], mods[k])
    mods["xt.model.tf_compat"].tf = tf
    mods["xt.model"].XTModel = type("XTModel", (object,), {})
    cfg = {}
    with open(os.path.join(REF, "xt/model/impala/default_config.py")) as f:
        exec(f.read(), cfg)                      # the reference's own GAMMA / LR
    mods["xt.model.impala.default_config"].GAMMA = cfg["GAMMA"]
    mods["xt.model.impala.default_config"].LR = cfg["LR"]
    sys.modules.update(mods)
    try:
        ppo_loss = _load("_ref_ppo_loss", "xt/model/ppo/__init__.py")
        tf_dist = _load("xt.model.tf_dist", "xt/model/tf_dist.py")
        vtrace = _load("xt.model.impala.vtrace", "xt/model/impala/vtrace.py")
        mods["xt.model.impala"].vtrace = vtrace
        opt = _load("_ref_impala_cnn_opt", "xt/model/impala/impala_cnn_opt.py")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ppo_loss, tf_dist, vtrace, opt, cfg["GAMMA"]


def wiring_source():
    """The lines of ImpalaCnnOpt.create_model from ``batch_step = self.sample_batch_steps`` to the end of
    the ``self.loss = vtrace_loss(...)`` statement, verbatim (impala_cnn_opt.py:169-196)."""
    with open(os.path.join(REF, "xt/model/impala/impala_cnn_opt.py")) as f:
        lines = f.read().split("\n")
    start = next(i for i, s in enumerate(lines) if s.strip() == "batch_step = self.sample_batch_steps")
    loss0 = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("self.loss = vtrace_loss("))
    end = next(i for i in range(loss0, len(lines)) if lines[i].strip() == ")")
    assert (start + 1, end + 1) == (169, 196), (start + 1, end + 1)
    return textwrap.dedent("\n".join(lines[start:end + 1]))


def f64(x):
    return np.asarray(x.detach().numpy() if isinstance(x, torch.Tensor) else x, np.float64)


# ----------------------------------------------------------------------------------------------
def ppo_categorical_case(ppo_loss, tf_dist, rng, b, a_dim, clip, entc, vfc, cc, big_v=False):
    logits = (rng.standard_normal((b, a_dim)) * 1.5).astype(np.float32)
    value = (rng.standard_normal((b, 1)) * (6.0 if big_v else 1.0)).astype(np.float32)
    action = rng.integers(0, a_dim, b).astype(np.int32)
    old_v = (value + rng.standard_normal((b, 1)) * (6.0 if big_v else 0.7)).astype(np.float32)
    target_v = (old_v + rng.standard_normal((b, 1)) * 2).astype(np.float32)
    adv = rng.standard_normal((b, 1)).astype(np.float32)
    # the behaviour log-prob: own log-prob + noise, so that ratios fall inside, on both sides outside of,
    # and (rows 0-3) EXACTLY on 1 (old_logp == the float64 log-prob the reference formula computes)
    lg = logits.astype(np.float64)
    own = (lg - lg.max(-1, keepdims=True))
    own = own - np.log(np.exp(own).sum(-1, keepdims=True))
    own = np.take_along_axis(own, action.reshape(-1, 1).astype(np.int64), 1)
    old_logp = (own + rng.standard_normal((b, 1)) * 0.15).astype(np.float32).astype(np.float64)
    old_logp[:4] = own[:4]                                       # float64 on purpose: exact ties
    # rows 4-7: |v - old_v| exactly VF_CLIP (binary-exact numbers), both signs
    for i, sgn in zip(range(4, 8), (1, -1, 1, -1)):
        old_v[i, 0] = np.float32(0.25 * i)
        value[i, 0] = np.float32(0.25 * i + sgn * vfc)
    tl = torch.tensor(logits.astype(np.float64), requires_grad=True).as_subclass(tf_shim.T)
    tv = torch.tensor(value.astype(np.float64), requires_grad=True).as_subclass(tf_shim.T)
    tl.retain_grad(); tv.retain_grad()
    dist = tf_dist.make_dist("Categorical", a_dim)
    dist.init_by_param(tl)
    t = lambda x: torch.tensor(np.asarray(x, np.float64)).as_subclass(tf_shim.T)
    actor = ppo_loss.actor_loss_with_entropy(dist, t(adv), t(old_logp), torch.tensor(action.astype(np.int64)),
                                             clip, entc)
    critic = ppo_loss.critic_loss(t(target_v), tv, t(old_v), vfc)
    loss = actor + cc * critic                                   # xt/model/ppo/ppo.py:89-92
    dl, dv = torch.autograd.grad(loss, [tl, tv])
    return dict(logits=logits, value=value, action=action, old_logp=old_logp, adv=adv, old_v=old_v,
                target_v=target_v, clip=clip, ent_coef=entc, vf_clip=vfc, critic_coef=cc,
                loss=f64(loss), actor_loss=f64(actor), critic_loss=f64(critic), dlogits=f64(dl), dvalue=f64(dv),
                logp=f64(dist.log_prob(torch.tensor(action.astype(np.int64)))), entropy=f64(dist.entropy()))


def ppo_gauss_case(ppo_loss, tf_dist, rng, b, a_dim, clip, entc, vfc, cc):
    mean = rng.standard_normal((b, a_dim)).astype(np.float32)
    log_std = (rng.standard_normal((1, a_dim)) * 0.3).astype(np.float32)
    value = rng.standard_normal((b, 1)).astype(np.float32)
    action = (mean + np.exp(log_std) * rng.standard_normal((b, a_dim))).astype(np.float32)
    old_v = (value + rng.standard_normal((b, 1))).astype(np.float32)
    target_v = (old_v + rng.standard_normal((b, 1)) * 2).astype(np.float32)
    adv = rng.standard_normal((b, 1)).astype(np.float32)
    t = lambda x: torch.tensor(np.asarray(x, np.float64)).as_subclass(tf_shim.T)
    tm = torch.tensor(mean.astype(np.float64), requires_grad=True).as_subclass(tf_shim.T)
    ts = torch.tensor(log_std.astype(np.float64), requires_grad=True).as_subclass(tf_shim.T)
    tv = torch.tensor(value.astype(np.float64), requires_grad=True).as_subclass(tf_shim.T)
    tf = sys.modules["oracle.tf_shim"].make_tf()
    dist = tf_dist.make_dist("DiagGaussian", a_dim)
    dist.init_by_param(tf.concat([tm, tm * 0.0 + ts], axis=-1))   # xt/model/ppo/ppo.py:75-79
    with torch.no_grad():
        own = f64(dist.log_prob(t(action)))
    old_logp = (own + rng.standard_normal((b, 1)) * 0.15).astype(np.float32).astype(np.float64)
    actor = ppo_loss.actor_loss_with_entropy(dist, t(adv), t(old_logp), t(action), clip, entc)
    critic = ppo_loss.critic_loss(t(target_v), tv, t(old_v), vfc)
    loss = actor + cc * critic
    dm, ds, dv = torch.autograd.grad(loss, [tm, ts, tv])
    return dict(mean=mean, log_std=log_std, value=value, action=action, old_logp=old_logp, adv=adv, old_v=old_v,
                target_v=target_v, clip=clip, ent_coef=entc, vf_clip=vfc, critic_coef=cc,
                loss=f64(loss), actor_loss=f64(actor), critic_loss=f64(critic), dmean=f64(dm), dlog_std=f64(ds),
                dvalue=f64(dv), logp=own, entropy=f64(dist.entropy()))


def impala_case(opt, vtrace, gamma, wiring, tf, rng, tlen, n_traj, a_dim, done_mode):
    """Flat env-major batch [n_traj*tlen] exactly as ImpalaCnnOpt.train feeds it."""
    n = tlen * n_traj
    logits = rng.standard_normal((n, a_dim)).astype(np.float32)
    baseline = rng.standard_normal(n).astype(np.float32)
    bp = (logits + rng.standard_normal((n, a_dim)) * 0.7).astype(np.float32)
    act = rng.integers(0, a_dim, n).astype(np.int32)
    rew = (rng.standard_normal(n) * 2).astype(np.float32)       # beyond [-1, 1]: clipped in the graph (:193)
    done = rng.random(n) < 0.1
    d2 = done.reshape(n_traj, tlen)
    if done_mode == "edges":                                     # done at t=0 and at t=T-2 (last step used)
        d2[:, 0] = True
        if tlen >= 2:
            d2[:, tlen - 2] = True
    elif done_mode == "none":
        d2[:] = False
    elif done_mode == "all":
        d2[:] = True
    done = d2.reshape(n)
    tl = torch.tensor(logits.astype(np.float64), requires_grad=True).as_subclass(tf_shim.T)
    tb = torch.tensor(baseline.astype(np.float64), requires_grad=True).as_subclass(tf_shim.T)
    holder = types.SimpleNamespace(
        sample_batch_steps=tlen, ph_bp_logic_outs=torch.tensor(bp.astype(np.float64)).as_subclass(tf_shim.T),
        pi_logic_outs=tl, ph_actions=torch.tensor(act.astype(np.int64)).as_subclass(tf_shim.T),
        ph_dones=torch.tensor(done).as_subclass(tf_shim.T),
        ph_rewards=torch.tensor(rew.astype(np.float64)).as_subclass(tf_shim.T), baseline=tb)
    ns = {"tf": tf, "self": holder, "vtrace_loss": opt.vtrace_loss, "GAMMA": gamma}
    exec(wiring, ns)                                             # impala_cnn_opt.py:169-196, verbatim
    loss = holder.loss
    dl, db = torch.autograd.grad(loss, [tl, tb])
    # the v-trace targets themselves, on the [T-1, B] views the wiring builds
    sb = ns["split_batches"]
    vs, pg = vtrace.from_logic_outputs(
        behaviour_policy_logic_outputs=sb(holder.ph_bp_logic_outs, drop_last=True),
        target_policy_logic_outputs=sb(tl.detach(), drop_last=True), actions=sb(holder.ph_actions, drop_last=True),
        discounts=sb(tf.cast(~holder.ph_dones, tf.float32) * gamma, drop_last=True),
        rewards=sb(tf.clip_by_value(holder.ph_rewards, -1, 1), drop_last=True),
        values=sb(tb.detach(), drop_last=True), bootstrap_value=sb(tb.detach())[-1])
    return dict(logits=logits, baseline=baseline, bp_logits=bp, actions=act, dones=done, rewards=rew,
                batch_step=tlen, gamma=gamma, loss=f64(loss), dlogits=f64(dl), dbaseline=f64(db),
                vs=f64(vs), pg_adv=f64(pg))


class KerasBackend(object):
    """The three ``K.`` functions the Keras-form loss closures use, over torch float64."""

    @staticmethod
    def log(x):
        return torch.log(x)

    @staticmethod
    def mean(x, axis=None):
        return torch.mean(x) if axis is None else torch.mean(x, dim=axis)

    @staticmethod
    def cast(x, dtype=None):
        return x.to(torch.float64)


def keras_loss_source(relpath, first, last):
    """``def impala_loss(advantage): ... return loss`` of ``relpath`` verbatim; asserts the line span."""
    with open(os.path.join(REF, relpath)) as f:
        lines = f.read().split("\n")
    start = next(i for i, s in enumerate(lines) if s.startswith("def impala_loss(advantage):"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip() == "return loss")
    assert (start + 1, end + 1) == (first, last), (relpath, start + 1, end + 1)
    return "\n".join(lines[start:end + 1])


def keras_impala_case(source, ent_coef, rng, b, a_dim, peaked):
    scope = {"K": KerasBackend, "ENTROPY_LOSS": ent_coef}
    exec(source, scope)
    logits = (rng.standard_normal((b, a_dim)) * (9.0 if peaked else 1.5)).astype(np.float32)
    if peaked:
        logits[:4, 0] += 40.0                    # p underflows next to 1e-10: the epsilon decides the value
    value = rng.standard_normal((b, 1)).astype(np.float32)
    adv = (rng.standard_normal((b, 1)) * 2).astype(np.float32)
    action = rng.integers(0, a_dim, b)
    onehot = np.eye(a_dim, dtype=np.float32)[action]
    target_v = (value + rng.standard_normal((b, 1))).astype(np.float32)
    lg = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    v = torch.tensor(value, dtype=torch.float64, requires_grad=True)
    policy = torch.softmax(lg, dim=-1)                       # the model's 'output_actions' activation
    per_sample = scope["impala_loss"](torch.tensor(adv, dtype=torch.float64))(
        torch.tensor(onehot, dtype=torch.float64), policy)   # <- the reference closure
    l_pi = per_sample.mean()                                 # Keras: batch mean of whatever the closure returns
    l_v = ((v - torch.tensor(target_v, dtype=torch.float64)) ** 2).mean()      # 'mse'
    loss = 1.0 * l_pi + 0.5 * l_v                            # loss_weights
    dl, dv = torch.autograd.grad(loss, [lg, v])
    return dict(logits=logits, value=value, adv=adv, onehot=onehot, target_v=target_v, ent_coef=ent_coef,
                per_sample=f64(per_sample).reshape(-1), loss=f64(loss), loss_pi=f64(l_pi), loss_v=f64(l_v),
                dlogits=f64(dl), dvalue=f64(dv))


def main():
    tf = tf_shim.make_tf()
    ppo_loss, tf_dist, vtrace, opt, gamma = load_reference(tf)
    wiring = wiring_source()
    os.makedirs(OUT, exist_ok=True)
    k = 0
    # PPO, categorical: breakout_ppo.yaml hyper-parameters (clip .1, ent .003, VF_CLIP 5 default, coef 1)
    # and an off-default set; A in {2 (CartPole), 4 (Breakout), 6 (Pong), 18 (full Atari)}
    for a_dim, b, hp, big in [(2, 64, (0.2, 0.01, 10.0, 0.5), False), (4, 320, (0.1, 0.003, 5.0, 1.0), True),
                              (6, 96, (0.1, 0.003, 0.5, 0.7), False), (18, 40, (0.3, 0.0, 5.0, 1.0), True)]:
        rng = np.random.default_rng(100 + k); k += 1
        case = ppo_categorical_case(ppo_loss, tf_dist, rng, b, a_dim, *hp, big_v=big)
        np.savez(os.path.join(OUT, "tf_ppo_cat_A{}_B{}.npz".format(a_dim, b)), **case)
        print("tf_ppo_cat", a_dim, b, float(case["loss"]))
    for a_dim, b in [(1, 48), (3, 200), (6, 33)]:               # pendulum A=1; odd sizes
        rng = np.random.default_rng(200 + k); k += 1
        case = ppo_gauss_case(ppo_loss, tf_dist, rng, b, a_dim, 0.2, 0.01, 10.0, 0.5)
        np.savez(os.path.join(OUT, "tf_ppo_gauss_A{}_B{}.npz".format(a_dim, b)), **case)
        print("tf_ppo_gauss", a_dim, b, float(case["loss"]))
    # IMPALA: breakout_impala.yaml (T=128, 1 and 4 trajectories), pong_impala_speedup (T=50, 20 traj, A=6),
    # degenerate T=2, A=18, done at t=0 and t=T-2, no done, all done
    for tlen, n_traj, a_dim, mode in [(128, 1, 4, "bernoulli"), (128, 4, 4, "edges"), (50, 20, 6, "bernoulli"),
                                      (2, 1, 4, "bernoulli"), (2, 3, 4, "all"), (5, 2, 18, "edges"),
                                      (50, 5, 6, "none"), (16, 3, 4, "all")]:
        rng = np.random.default_rng(300 + k); k += 1
        case = impala_case(opt, vtrace, gamma, wiring, tf, rng, tlen, n_traj, a_dim, mode)
        np.savez(os.path.join(OUT, "tf_impala_T{}_B{}_A{}_{}.npz".format(tlen, n_traj, a_dim, mode)), **case)
        print("tf_impala", tlen, n_traj, a_dim, mode, float(case["loss"]))
    # Keras-form IMPALA loss (ImpalaCnn / ImpalaMlp): both closures, plain and peaked policies
    cfg = {}
    with open(os.path.join(REF, "xt/model/impala/default_config.py")) as f:
        exec(f.read(), cfg)
    for tag, rel, span in [("cnn", "xt/model/impala/impala_cnn.py", (99, 108)),
                           ("mlp", "xt/model/impala/impala_mlp.py", (84, 93))]:
        src = keras_loss_source(rel, *span)
        for a_dim, b, peaked in [(4, 128, False), (6, 50, True), (2, 33, False)]:
            rng = np.random.default_rng(400 + k); k += 1
            case = keras_impala_case(src, cfg["ENTROPY_LOSS"], rng, b, a_dim, peaked)
            np.savez(os.path.join(OUT, "tf_keras_impala_{}_A{}_B{}.npz".format(tag, a_dim, b)), **case)
            print("tf_keras_impala", tag, a_dim, b, float(case["loss"]))


if __name__ == "__main__":
    main()
