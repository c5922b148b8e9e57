"""Oracle (test infrastructure): float64 numpy restatement of the TF graphs on the hot path.

The reference builds these with Keras/TensorFlow (an un-vendored dependency), so
this file restates TF's published semantics; every function cites the reference
call site it stands in for.  Layouts are TensorFlow's: activations NHWC, conv
kernels HWIO ``[kh,kw,cin,cout]``, dense kernels ``[in,out]``, Flatten in
(H,W,C) order.  Parameter dict keys are the TF variable names produced by the
reference's Keras layer names (xt/model/model_utils.py:34-36,87,96).

Parity unpinned at the TF boundary (see oracle/__init__.py); gradients here are
hand-derived and cross-checked against torch autograd in tests/test_oracle.py.
"""
from collections import OrderedDict

import numpy as np


# ----------------------------------------------------------------------------
# geometry
# ----------------------------------------------------------------------------
def conv_out_size(size, k, s, padding):
    """TF output size + (pad_before, pad_after).  SAME pads the extra cell after."""
    if padding == "valid":
        return (size - k) // s + 1, 0, 0
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    before = total // 2
    return out, before, total - before


class LayerSpec(object):
    """One Conv2D / Dense layer of a trunk."""

    def __init__(self, name, kind, cin, cout, act, k=1, s=1, padding="valid", in_hw=(1, 1)):
        self.name, self.kind, self.cin, self.cout, self.act = name, kind, cin, cout, act
        self.k, self.s, self.padding = k, s, padding
        self.in_h, self.in_w = in_hw
        if kind == "conv":
            self.out_h, self.pt, self.pb = conv_out_size(self.in_h, k, s, padding)
            self.out_w, self.pl, self.pr = conv_out_size(self.in_w, k, s, padding)
        else:
            self.out_h = self.out_w = 1
            self.pt = self.pb = self.pl = self.pr = 0

    @property
    def kernel_shape(self):
        if self.kind == "conv":
            return (self.k, self.k, self.cin, self.cout)
        return (self.cin, self.cout)


def ppo_cnn_filters(state_dim):
    """xt/model/model_utils.py:120-149 (``get_default_filters``), (cout, k, s)."""
    hw = list(state_dim[:2])
    if hw == [84, 84]:
        return [(32, 8, 4), (32, 4, 2), (64, 3, 1)]
    if hw == [42, 42]:
        return [(32, 4, 2), (32, 4, 2), (64, 3, 1)]
    if hw == [15, 15]:
        return [(32, 5, 1), (64, 3, 1), (64, 3, 1)]
    if len(state_dim) != 3 or hw[0] != hw[1] or hw[0] > 64:
        raise ValueError("no default filters for %r" % (state_dim,))
    # inferred architecture, model_utils.py:150-176 (sizes <= 64: beyond that the rule's kernel exceeds the image)
    filters, size, flat, n = [], hw[0], False, 16
    while not flat:
        if size <= 3:
            k, s, flat = 1, 1, True
        elif size <= 8:
            k, s, flat = 3, 1, True
        else:
            k, s, flat = 5, 2, False
        filters.append((n, k, s))
        n *= 2
        size //= s
    return filters


def impala_filters(state_dim):
    """xt/model/atari_model.py:4-23 (``get_atari_filter``)."""
    hw = list(state_dim[:2])
    if hw == [84, 84]:
        return [(16, 8, 4), (32, 4, 2), (256, 11, 1)]
    if hw == [42, 42]:
        return [(16, 4, 2), (32, 4, 2), (256, 11, 1)]
    raise ValueError("no default filters for %r" % (state_dim,))


def _conv_trunk(prefix_fmt, state_dim, filters, act, paddings):
    h, w, c = state_dim
    layers = []
    for i, (cout, k, s) in enumerate(filters):
        spec = LayerSpec(prefix_fmt.format(i), "conv", c, cout, act, k, s, paddings[i], (h, w))
        layers.append(spec)
        h, w, c = spec.out_h, spec.out_w, cout
    return layers, h * w * c


def _mlp_trunk(prefix, in_dim, hidden_sizes, act):
    layers = []
    for i, hsz in enumerate(hidden_sizes):
        layers.append(LayerSpec("{}_hidden_mlp_{}".format(prefix, i), "dense", in_dim, hsz, act))
        in_dim = hsz
    return layers, in_dim


def ppo_cnn_spec(state_dim, action_dim, hidden_sizes=(512,), act="relu", vf_share=True):
    """Layer lists for ``get_cnn_backbone`` xt/model/model_utils.py:49-80."""
    trunks = []
    for prefix in (["shared"] if vf_share else ["pi", "v"]):
        convs, flat = _conv_trunk(prefix + "_conv_layer_{}", state_dim, ppo_cnn_filters(state_dim),
                                  act, ["valid"] * 3)
        mlps, feat = _mlp_trunk(prefix, flat, hidden_sizes, act)
        trunks.append(convs + mlps)
    return dict(trunks=trunks, feat=feat, action_dim=action_dim, pi_name="pi_latent",
                v_name="output_value", input_scale=("div", 0.0, 255.0))


def ppo_mlp_spec(state_dim, action_dim, hidden_sizes=(64, 64), act="tanh", vf_share=False,
                 action_type="Categorical"):
    """Layer lists for ``get_mlp_backbone`` xt/model/model_utils.py:22-46.  ``action_type`` 'DiagGaussian'
    adds the state-independent ``pi_logstd`` [1,A] variable of xt/model/ppo/ppo.py:75-79."""
    trunks = []
    for prefix in (["shared"] if vf_share else ["pi", "v"]):
        mlps, feat = _mlp_trunk(prefix, int(state_dim[0]), hidden_sizes, act)
        trunks.append(mlps)
    return dict(trunks=trunks, feat=feat, action_dim=action_dim, pi_name="pi_latent",
                v_name="output_value", input_scale=None, action_type=action_type)


def impala_cnn_opt_spec(state_dim, action_dim, state_mean=0.0, state_std=255.0):
    """Layer list for ``ImpalaCnnOpt.create_model`` impala_cnn_opt.py:110-152.

    Keras auto-names the Conv2D layers conv2d, conv2d_1, conv2d_2 (trunk) and
    conv2d_3 (1x1 policy head) under scope ``explore_agent``; the baseline is
    ``tf.layers.dense`` -> ``dense``.
    """
    filt = impala_filters(state_dim)
    names = ["explore_agent/conv2d", "explore_agent/conv2d_1", "explore_agent/conv2d_2"]
    h, w, c = state_dim
    layers = []
    for i, (cout, k, s) in enumerate(filt):
        pad = "same" if i < len(filt) - 1 else "valid"
        spec = LayerSpec(names[i], "conv", c, cout, "relu", k, s, pad, (h, w))
        layers.append(spec)
        h, w, c = spec.out_h, spec.out_w, cout
    assert (h, w) == (1, 1)
    return dict(trunks=[layers], feat=c, action_dim=action_dim, pi_name="explore_agent/conv2d_3",
                v_name="explore_agent/dense",
                input_scale=("affine", float(state_mean), float(state_std)))


def impala_cnn_spec(state_dim, action_dim):
    """Keras ``ImpalaCnn.create_model`` (xt/model/impala/impala_cnn.py:44-57): Conv2D 32@8x8/4, 64@4x4/2, 64@3x3/1
    (valid, relu) on uint8/255, Dense 256 relu, then a SOFTMAX policy head ``output_actions`` and ``output_value`` on
    the same features.  Keras auto-names: conv2d, conv2d_1, conv2d_2, dense."""
    filt = [(32, 8, 4), (64, 4, 2), (64, 3, 1)]
    names = ["conv2d", "conv2d_1", "conv2d_2"]
    h, w, c = state_dim
    layers = []
    for (cout, k, s), name in zip(filt, names):
        spec = LayerSpec(name, "conv", c, cout, "relu", k, s, "valid", (h, w))
        layers.append(spec)
        h, w, c = spec.out_h, spec.out_w, cout
    layers.append(LayerSpec("dense", "dense", h * w * c, 256, "relu"))
    return dict(trunks=[layers], feat=256, action_dim=action_dim, pi_name="output_actions", v_name="output_value",
                input_scale=("div", 0.0, 255.0))


def impala_mlp_spec(state_dim, action_dim, hidden_size=128, num_layers=1):
    """Keras ``ImpalaMlp.create_model`` (xt/model/impala/impala_mlp.py:42-53): NUM_LAYERS x Dense(HIDDEN_SIZE, relu)
    named dense, dense_1, ...; softmax policy + value heads."""
    layers, cin = [], int(state_dim[0])
    for i in range(num_layers):
        layers.append(LayerSpec("dense" if i == 0 else "dense_%d" % i, "dense", cin, hidden_size, "relu"))
        cin = hidden_size
    return dict(trunks=[layers], feat=hidden_size, action_dim=action_dim, pi_name="output_actions",
                v_name="output_value", input_scale=None)


# ----------------------------------------------------------------------------
# initialisers (Keras defaults) -- only used to make seeded test weights
# ----------------------------------------------------------------------------
def glorot_uniform(rng, shape, dtype=np.float32):
    if len(shape) == 4:
        rf = shape[0] * shape[1]
        fan_in, fan_out = rf * shape[2], rf * shape[3]
    else:
        fan_in, fan_out = shape
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def init_params(spec, seed=0, dtype=np.float32, bias_scale=0.0):
    """OrderedDict of TF-named variables in TF creation order."""
    rng = np.random.default_rng(seed)
    params = OrderedDict()

    def add(name, shape):
        params[name + "/kernel"] = glorot_uniform(rng, shape, dtype)
        b = np.zeros((shape[-1],), dtype)
        if bias_scale:
            b = (rng.standard_normal(shape[-1]) * bias_scale).astype(dtype)
        params[name + "/bias"] = b

    n_tr = len(spec["trunks"])
    if n_tr == 1:
        for lay in spec["trunks"][0]:
            add(lay.name, lay.kernel_shape)
        add(spec["pi_name"], (spec["feat"], spec["action_dim"]))
        add(spec["v_name"], (spec["feat"], 1))
    else:
        # get_mlp_backbone: pi trunk, pi_latent, v trunk, output_value (model_utils.py:32-36);
        # get_cnn_backbone unshared: pi convs, v convs, pi mlp, v mlp, heads (:67-74)
        pi, v = spec["trunks"]
        if pi[0].kind == "dense":
            for lay in pi:
                add(lay.name, lay.kernel_shape)
            add(spec["pi_name"], (spec["feat"], spec["action_dim"]))
            for lay in v:
                add(lay.name, lay.kernel_shape)
            add(spec["v_name"], (spec["feat"], 1))
        else:
            for grp in ("conv", "dense"):
                for tr in (pi, v):
                    for lay in tr:
                        if lay.kind == grp:
                            add(lay.name, lay.kernel_shape)
            add(spec["pi_name"], (spec["feat"], spec["action_dim"]))
            add(spec["v_name"], (spec["feat"], 1))
    if spec.get("action_type") == "DiagGaussian":
        # tf.get_variable('pi_logstd', (1, A), zeros) is created in build_graph, after the Keras model (ppo.py:78)
        params["pi_logstd"] = np.zeros((1, spec["action_dim"]), dtype)
    return params


# ----------------------------------------------------------------------------
# layer math
# ----------------------------------------------------------------------------
def im2col(x, lay):
    """x [B,H,W,C] -> cols [B*OH*OW, k*k*C] with TF padding (zeros)."""
    b = x.shape[0]
    xp = np.pad(x, ((0, 0), (lay.pt, lay.pb), (lay.pl, lay.pr), (0, 0)))
    k, s = lay.k, lay.s
    cols = np.empty((b, lay.out_h, lay.out_w, k, k, lay.cin), x.dtype)
    for ky in range(k):
        for kx in range(k):
            cols[:, :, :, ky, kx, :] = xp[:, ky:ky + s * lay.out_h:s, kx:kx + s * lay.out_w:s, :]
    return cols.reshape(b * lay.out_h * lay.out_w, k * k * lay.cin)


def col2im(dcols, lay, b):
    k, s = lay.k, lay.s
    dcols = dcols.reshape(b, lay.out_h, lay.out_w, k, k, lay.cin)
    dxp = np.zeros((b, lay.in_h + lay.pt + lay.pb, lay.in_w + lay.pl + lay.pr, lay.cin), dcols.dtype)
    for ky in range(k):
        for kx in range(k):
            dxp[:, ky:ky + s * lay.out_h:s, kx:kx + s * lay.out_w:s, :] += dcols[:, :, :, ky, kx, :]
    return dxp[:, lay.pt:lay.pt + lay.in_h, lay.pl:lay.pl + lay.in_w, :]


SELU_SCALE, SELU_ALPHA, LEAKY_ALPHA = 1.0507009873554805, 1.6732632423543772, 0.2     # tf.nn.selu / tf.nn.leaky_relu


def act_fwd(z, act):
    """ACTIVATION_MAP of xt/model/model_utils.py:8-20 (TF 1.15 op definitions); swish / gelu are not supported by the
    product (their derivative needs the pre-activation, which is not stored) and are absent here too."""
    if act == "relu":
        return np.maximum(z, 0)
    if act == "tanh":
        return np.tanh(z)
    if act == "sigmoid":
        return 1.0 / (1.0 + np.exp(-z))
    if act == "softsign":
        return z / (1.0 + np.abs(z))
    if act == "softplus":
        return np.logaddexp(0.0, z)
    if act == "leaky_relu":
        return np.where(z > 0, z, LEAKY_ALPHA * z)
    if act == "elu":
        return np.where(z > 0, z, np.expm1(np.minimum(z, 0)))
    if act == "selu":
        return SELU_SCALE * np.where(z > 0, z, SELU_ALPHA * np.expm1(np.minimum(z, 0)))
    if act == "swish":                  # tf.nn.swish: x * sigmoid(x)
        return z / (1.0 + np.exp(-z))
    if act == "gelu":                   # xt/model/tf_utils.py:157-166 (the tanh form)
        return 0.5 * z * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (z + 0.044715 * z ** 3)))
    if act is None or act == "none":
        return z
    raise KeyError(act)


NEEDS_PREACT = ("swish", "gelu")        # not monotonic: the derivative needs the pre-activation


def act_bwd(dy, y, act, z=None):
    """d(pre-activation) from d(post) and the saved OUTPUT y.  The pre-activation z is recovered from y in closed
    form (every supported activation is strictly monotonic) and the textbook derivative is evaluated at z -- not the
    from-the-output shortcuts the kernels use (cross-checked against torch autograd in tests/test_oracle.py)."""
    if act == "relu":
        return dy * (y > 0)
    if act == "tanh":
        return dy * (1.0 - y * y)
    if act == "sigmoid":
        return dy * y * (1.0 - y)
    if act == "softsign":
        z = y / (1.0 - np.abs(y))
        return dy / np.square(1.0 + np.abs(z))
    if act == "softplus":
        z = y + np.log(-np.expm1(-y))                # y = log(1 + e^z)  ->  z = log(e^y - 1)
        return dy / (1.0 + np.exp(-z))
    if act == "leaky_relu":
        return dy * np.where(y > 0, 1.0, LEAKY_ALPHA)
    if act == "elu":
        z = np.where(y > 0, y, np.log1p(np.minimum(y, 0)))
        return dy * np.where(z > 0, 1.0, np.exp(np.minimum(z, 0)))
    if act == "selu":
        z = np.where(y > 0, y / SELU_SCALE, np.log1p(np.minimum(y, 0) / (SELU_SCALE * SELU_ALPHA)))
        return dy * SELU_SCALE * np.where(z > 0, 1.0, SELU_ALPHA * np.exp(np.minimum(z, 0)))
    if act == "swish":
        sg = 1.0 / (1.0 + np.exp(-z))
        return dy * (sg + z * sg * (1.0 - sg))
    if act == "gelu":
        c = np.sqrt(2.0 / np.pi)
        th = np.tanh(c * (z + 0.044715 * z ** 3))
        return dy * (0.5 * (1.0 + th) + 0.5 * z * (1.0 - th * th) * c * (1.0 + 3 * 0.044715 * z * z))
    if act is None or act == "none":
        return dy
    raise KeyError(act)


class ActorCritic(object):
    """Forward/backward of the pi/v network described by a spec, in ``dtype``."""

    def __init__(self, spec, params, dtype=np.float64):
        self.spec, self.dtype = spec, dtype
        self.params = OrderedDict((k, np.asarray(v, dtype)) for k, v in params.items())

    def _transform(self, obs):
        sc = self.spec["input_scale"]
        x = np.asarray(obs).astype(self.dtype)
        if sc is None or np.asarray(obs).dtype != np.uint8:
            return x
        kind, mean, std = sc
        if kind == "div":  # layer_function, model_utils.py:187-189
            return x / self.dtype(std)
        # state_transform, model_utils.py:192-201
        if abs(mean) < 1e-4:
            return x / self.dtype(std)
        return (x - self.dtype(mean)) / self.dtype(std)

    def forward(self, obs):
        x0 = self._transform(obs)
        b = x0.shape[0]
        self.cache = []
        feats = []
        for trunk in self.spec["trunks"]:
            x = x0
            tc = []
            for lay in trunk:
                w = self.params[lay.name + "/kernel"]
                bias = self.params[lay.name + "/bias"]
                if lay.kind == "conv":
                    cols = im2col(x.reshape(b, lay.in_h, lay.in_w, lay.cin), lay)
                    z = cols @ w.reshape(-1, lay.cout) + bias
                    y = act_fwd(z, lay.act).reshape(b, lay.out_h, lay.out_w, lay.cout)
                    z = z.reshape(y.shape)
                else:
                    cols = x.reshape(b, -1)
                    z = cols @ w + bias
                    y = act_fwd(z, lay.act)
                tc.append((cols, y, z))
                x = y
            self.cache.append(tc)
            feats.append(x.reshape(b, -1))
        self.feats = feats
        f_pi, f_v = feats[0], feats[-1]
        logits = f_pi @ self.params[self.spec["pi_name"] + "/kernel"].reshape(f_pi.shape[1], -1) \
            + self.params[self.spec["pi_name"] + "/bias"]
        value = f_v @ self.params[self.spec["v_name"] + "/kernel"] + self.params[self.spec["v_name"] + "/bias"]
        return logits, value

    def backward(self, dlogits, dvalue, extra=None):
        """dlogits [B,A], dvalue [B,1] -> grads dict (same keys as params).  ``extra``: gradients of variables
        outside the layer stack (pi_logstd)."""
        grads = OrderedDict(extra or {})
        f_pi, f_v = self.feats[0], self.feats[-1]
        wpi = self.params[self.spec["pi_name"] + "/kernel"]
        wv = self.params[self.spec["v_name"] + "/kernel"]
        grads[self.spec["pi_name"] + "/kernel"] = (f_pi.T @ dlogits).reshape(wpi.shape)
        grads[self.spec["pi_name"] + "/bias"] = dlogits.sum(0)
        grads[self.spec["v_name"] + "/kernel"] = f_v.T @ dvalue
        grads[self.spec["v_name"] + "/bias"] = dvalue.sum(0)
        dfeat = [None] * len(self.spec["trunks"])
        dfeat[0] = dlogits @ wpi.reshape(f_pi.shape[1], -1).T
        dv_feat = dvalue @ wv.T
        if len(dfeat) == 1:
            dfeat[0] = dfeat[0] + dv_feat
        else:
            dfeat[1] = dv_feat
        b = dlogits.shape[0]
        for trunk, tc, dy in zip(self.spec["trunks"], self.cache, dfeat):
            for li in range(len(trunk) - 1, -1, -1):
                lay = trunk[li]
                cols, y, z = tc[li]
                dz = act_bwd(dy.reshape(y.shape), y, lay.act, z)
                dz2 = dz.reshape(-1, lay.cout)
                w = self.params[lay.name + "/kernel"]
                grads[lay.name + "/kernel"] = (cols.T @ dz2).reshape(w.shape)
                grads[lay.name + "/bias"] = dz2.sum(0)
                if li > 0:
                    dcols = dz2 @ w.reshape(-1, lay.cout).T
                    dy = col2im(dcols, lay, b) if lay.kind == "conv" else dcols
        return OrderedDict((k, grads[k]) for k in self.params)


# ----------------------------------------------------------------------------
# categorical distribution + PPO loss (xt/model/tf_dist.py:89-130, xt/model/ppo/__init__.py:4-25)
# ----------------------------------------------------------------------------
def softmax_stats(logits):
    m = logits.max(axis=-1, keepdims=True)
    rl = logits - m
    e = np.exp(rl)
    z = e.sum(axis=-1, keepdims=True)
    p = e / z
    logp_all = rl - np.log(z)
    ent = (p * (np.log(z) - rl)).sum(axis=-1, keepdims=True)  # tf_dist.py:108-113
    return p, logp_all, ent


def ppo_loss_and_grads(logits, value, action, old_logp, adv, old_v, target_v,
                       clip_ratio, ent_coef, vf_clip, critic_coef):
    """loss = actor_loss_with_entropy + critic_coef*critic_loss (xt/model/ppo/ppo.py:89-92).

    All label arrays are [B,1] (action [B]).  Returns (loss, dlogits [B,A], dvalue [B,1],
    parts dict).  Gradient conventions follow TF: min/max send the gradient to the
    first argument on ties, clip_by_value passes gradient on the closed interval.
    """
    dt = logits.dtype
    bsz = logits.shape[0]
    p, logp_all, ent = softmax_stats(logits)
    a = np.asarray(action).astype(np.int64).reshape(-1, 1)
    logp = np.take_along_axis(logp_all, a, axis=1)          # -neglog_prob, tf_dist.py:103-106
    ratio = np.exp(logp - old_logp)
    surr1 = ratio * adv
    clipped = np.clip(ratio, 1.0 - clip_ratio, 1.0 + clip_ratio)
    surr2 = clipped * adv
    surr = np.minimum(surr1, surr2)
    actor_loss = -surr.mean() - ent_coef * ent.mean()
    vf1 = np.square(value - target_v)
    vclip = old_v + np.clip(value - old_v, -vf_clip, vf_clip)
    vf2 = np.square(vclip - target_v)
    critic = 0.5 * np.maximum(vf1, vf2).mean()
    loss = actor_loss + critic_coef * critic

    # gradients
    first = surr1 <= surr2                       # tf.minimum: grad to x where x <= y
    in_rng = (ratio >= 1.0 - clip_ratio) & (ratio <= 1.0 + clip_ratio)
    dsurr_dratio = np.where(first, adv, np.where(in_rng, adv, 0.0))
    dlogp = -(dsurr_dratio * ratio) / bsz
    onehot = np.zeros_like(logits)
    np.put_along_axis(onehot, a, 1.0, axis=1)
    dlogits = dlogp * (onehot - p)
    dent_dlogits = -p * (logp_all + ent)
    dlogits = dlogits - (ent_coef / bsz) * dent_dlogits
    take1 = vf1 >= vf2                           # tf.maximum: grad to x where x >= y
    in_v = np.abs(value - old_v) <= vf_clip
    dv = np.where(take1, 2.0 * (value - target_v), np.where(in_v, 2.0 * (vclip - target_v), 0.0))
    dvalue = (critic_coef * 0.5 / bsz) * dv
    parts = dict(actor_loss=actor_loss, critic_loss=critic, entropy=ent.mean(), logp=logp)
    return dt.type(loss), dlogits.astype(dt), dvalue.astype(dt), parts


def gauss_ppo_loss_and_grads(mean, log_std, value, action, old_logp, adv, old_v, target_v,
                             clip_ratio, ent_coef, vf_clip, critic_coef):
    """PPO loss with ``DiagGaussianDist`` (xt/model/tf_dist.py:47-87; dist_param = concat([pi_latent,
    pi_latent*0 + pi_logstd]), xt/model/ppo/ppo.py:75-79).  mean [B,A], log_std [1,A], action [B,A] float.
    Returns (loss, dmean [B,A], dvalue [B,1], dlog_std [1,A], parts)."""
    dt = mean.dtype
    bsz, adim = mean.shape
    log_std = np.asarray(log_std, dt).reshape(1, adim)
    std = np.exp(log_std)
    x = np.asarray(action, dt).reshape(bsz, adim)
    zz = (x - mean) / std
    # neglog_prob, tf_dist.py:66-69 (the python-float constant is cast to the tensor dtype)
    neglogp = dt.type(0.5 * np.log(2.0 * np.pi)) * dt.type(adim) + 0.5 * np.square(zz).sum(-1, keepdims=True) \
        + (log_std + 0.0 * mean).sum(-1, keepdims=True)
    logp = -neglogp
    ent = (log_std + dt.type(0.5 * (np.log(2.0 * np.pi) + 1.0)) + 0.0 * mean).sum(-1, keepdims=True)   # :74-75
    ratio = np.exp(logp - old_logp)
    surr1 = ratio * adv
    clipped = np.clip(ratio, 1.0 - clip_ratio, 1.0 + clip_ratio)
    surr2 = clipped * adv
    surr = np.minimum(surr1, surr2)
    actor_loss = -surr.mean() - ent_coef * ent.mean()
    vf1 = np.square(value - target_v)
    vclip = old_v + np.clip(value - old_v, -vf_clip, vf_clip)
    vf2 = np.square(vclip - target_v)
    critic = 0.5 * np.maximum(vf1, vf2).mean()
    loss = actor_loss + critic_coef * critic
    first = surr1 <= surr2
    in_rng = (ratio >= 1.0 - clip_ratio) & (ratio <= 1.0 + clip_ratio)
    dsurr_dratio = np.where(first, adv, np.where(in_rng, adv, 0.0))
    dlogp = -(dsurr_dratio * ratio) / bsz
    dmean = dlogp * zz / std
    dls_rows = dlogp * (np.square(zz) - 1.0) - (ent_coef / bsz)
    dlog_std = dls_rows.sum(0, keepdims=True)
    take1 = vf1 >= vf2
    in_v = np.abs(value - old_v) <= vf_clip
    dv = np.where(take1, 2.0 * (value - target_v), np.where(in_v, 2.0 * (vclip - target_v), 0.0))
    dvalue = (critic_coef * 0.5 / bsz) * dv
    parts = dict(actor_loss=actor_loss, critic_loss=critic, entropy=ent.mean(), logp=logp, dls_rows=dls_rows)
    return dt.type(loss), dmean.astype(dt), dvalue.astype(dt), dlog_std.astype(dt), parts


# ----------------------------------------------------------------------------
# IMPALA v-trace loss (impala_cnn_opt.py:188-196, 299-351)
# ----------------------------------------------------------------------------
def impala_loss_and_grads(logits, baseline, bp_logits, actions, dones, rewards, batch_step,
                          gamma=0.99, dtype=np.float64):
    """Flat env-major inputs [B*T(,A)] -> (loss, dlogits [B*T,A], dbaseline [B*T]).

    pi_loss + 0.5*baseline_loss + 0.01*entropy_loss, all sums (impala_cnn_opt.py:299-351);
    the last time step of every trajectory is only the bootstrap (:188-196).
    """
    from oracle.returns import split_batches, vtrace_from_logits, sparse_softmax_ce
    logits = np.asarray(logits, dtype)
    baseline = np.asarray(baseline, dtype)
    tp = split_batches(logits, batch_step, drop_last=True)
    bp = split_batches(np.asarray(bp_logits, dtype), batch_step, drop_last=True)
    act = split_batches(np.asarray(actions), batch_step, drop_last=True)
    disc = split_batches((~np.asarray(dones, bool)).astype(dtype) * dtype(gamma), batch_step, drop_last=True)
    rew = split_batches(np.clip(np.asarray(rewards, dtype), -1, 1), batch_step, drop_last=True)
    vals = split_batches(baseline, batch_step, drop_last=True)
    boot = split_batches(baseline, batch_step)[-1]
    vs, pg_adv = vtrace_from_logits(bp, tp, act, disc, rew, vals, boot, dtype=dtype)
    ce = sparse_softmax_ce(tp, act)
    pi_loss = (ce * pg_adv).sum()
    val_loss = 0.5 * np.square(vs - vals).sum()
    p, logp_all, ent = softmax_stats(tp)
    entropy_loss = -(-(p * logp_all).sum(axis=-1)).sum()
    loss = pi_loss + 0.5 * val_loss + 0.01 * entropy_loss
    # grads (vs, pg_adv are stop_gradient)
    onehot = np.zeros_like(tp)
    np.put_along_axis(onehot, act[..., None].astype(np.int64), 1.0, axis=-1)
    d_tp = pg_adv[..., None] * (p - onehot)
    h = -(p * logp_all).sum(axis=-1, keepdims=True)
    d_tp = d_tp + 0.01 * (p * (logp_all + h))      # d(-H)/dlogits = p*(log p + H)
    d_vals = 0.5 * (vals - vs)
    tlen = batch_step
    bcount = logits.shape[0] // tlen
    dlogits = np.zeros((bcount, tlen, logits.shape[-1]), dtype)
    dlogits[:, :-1] = np.swapaxes(d_tp, 0, 1)
    dbase = np.zeros((bcount, tlen), dtype)
    dbase[:, :-1] = np.swapaxes(d_vals, 0, 1)
    return dtype(loss), dlogits.reshape(logits.shape), dbase.reshape(baseline.shape), dict(vs=vs, pg_adv=pg_adv)


# ----------------------------------------------------------------------------
# optimiser (tf.clip_by_global_norm + tf.train.AdamOptimizer, xt/model/ppo/ppo.py:97-102)
# ----------------------------------------------------------------------------
def clip_by_global_norm(grads, clip_norm):
    """grads dict -> (clipped dict, global_norm).  t*clip/max(norm, clip)."""
    dt = next(iter(grads.values())).dtype
    gn = np.sqrt(sum(np.square(g.astype(np.float64)).sum() for g in grads.values())).astype(dt)
    scale = dt.type(clip_norm) * min(dt.type(1.0) / gn if gn > 0 else np.inf, dt.type(1.0) / dt.type(clip_norm))
    return OrderedDict((k, g * scale) for k, g in grads.items()), gn


class AdamTF(object):
    """TF1 ``AdamOptimizer``: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps)."""

    def __init__(self, params, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m = OrderedDict((k, np.zeros_like(v)) for k, v in params.items())
        self.v = OrderedDict((k, np.zeros_like(v)) for k, v in params.items())

    def apply(self, params, grads):
        self.t += 1
        dt = next(iter(params.values())).dtype.type
        lr_t = dt(self.lr) * np.sqrt(dt(1.0) - dt(self.b2) ** self.t) / (dt(1.0) - dt(self.b1) ** self.t)
        for k in params:
            g = grads[k]
            self.m[k] += (g - self.m[k]) * dt(1.0 - self.b1)
            self.v[k] += (g * g - self.v[k]) * dt(1.0 - self.b2)
            params[k] -= lr_t * self.m[k] / (np.sqrt(self.v[k]) + dt(self.eps))


# ----------------------------------------------------------------------------
# whole updates
# ----------------------------------------------------------------------------
class RmsPropTF(object):
    """tf.train.RMSPropOptimizer(lr, decay, momentum=0, epsilon, centered=True) as TF1's
    apply_centered_rms_prop computes it (slots: rms initialised to ONES, mg and momentum to zeros):
    ms = d*ms + (1-d)*g^2; mg = d*mg + (1-d)*g; var -= lr*g / sqrt(ms - mg^2 + eps)
    (the reference's opt_type 'rmsprop', xt/model/impala/impala_cnn_opt.py:205-206)."""

    def __init__(self, params, lr, decay=0.99, eps=0.1):
        self.lr, self.decay, self.eps = lr, decay, eps
        self.ms = OrderedDict((k, np.ones_like(v)) for k, v in params.items())
        self.mg = OrderedDict((k, np.zeros_like(v)) for k, v in params.items())

    def apply(self, params, grads):
        d = self.decay
        for k in params:
            g = grads[k].astype(params[k].dtype)
            self.ms[k] = d * self.ms[k] + (1.0 - d) * g * g
            self.mg[k] = d * self.mg[k] + (1.0 - d) * g
            params[k] -= self.lr * g / np.sqrt(self.ms[k] - self.mg[k] * self.mg[k] + self.eps)


def keras_impala_loss_and_grads(logits, value, adv, onehot, target_v, ent_coef, eps=1e-10):
    """Loss of the non-opt IMPALA models (impala_cnn.py:94-108 / impala_mlp.py:83-93 + ``model.compile``):

        p = softmax(logits)
        L_pi = mean_{b,a}( adv_b * (-y_ba * log(p_ba + 1e-10)) - ENT * (-p_ba * log(p_ba + 1e-10)) )
        L_v  = mean_b (v_b - target_b)^2          (Keras 'mse')
        L    = 1.0 * L_pi + 0.5 * L_v             (loss_weights)

    -> (L, dL/dlogits [B,A], dL/dvalue [B], (L_pi, L_v)).  adv / target_v: [B] or [B,1]; onehot: [B,A]."""
    dt = logits.dtype
    b, a = logits.shape
    adv = np.asarray(adv, dt).reshape(b, 1)
    tv = np.asarray(target_v, dt).reshape(b)
    y = np.asarray(onehot, dt)
    z = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(z)
    p = e / e.sum(axis=1, keepdims=True)
    lp = np.log(p + dt.type(eps))
    per = adv * (-y * lp) + dt.type(ent_coef) * (p * lp)
    l_pi = per.mean()
    diff = value.reshape(b) - tv
    l_v = (diff * diff).mean()
    # d per / d p = -adv*y/(p+eps) + ENT*(log(p+eps) + p/(p+eps)), then through the softmax
    g = (-adv * y / (p + dt.type(eps)) + dt.type(ent_coef) * (lp + p / (p + dt.type(eps)))) / dt.type(b * a)
    dlogits = p * (g - (p * g).sum(axis=1, keepdims=True))
    dvalue = (dt.type(0.5) * dt.type(2.0) * diff / dt.type(b)).reshape(b, 1)
    return l_pi + dt.type(0.5) * l_v, dlogits, dvalue, (l_pi, l_v)


class AdamKeras(object):
    """``tf.keras.optimizers.Adam(lr, clipnorm=..., decay=...)`` (OptimizerV2, epsilon = 1e-7):
    every gradient TENSOR is clipped to ``clipnorm`` on its own (clip_by_norm, not the global norm);
    lr_t = lr / (1 + decay * iterations) * sqrt(1 - b2^t) / (1 - b1^t) with t = iterations + 1;
    m, v as usual; theta -= lr_t * m / (sqrt(v) + eps)."""

    def __init__(self, params, lr, clipnorm=None, decay=0.0, beta1=0.9, beta2=0.999, eps=1e-7):
        self.lr, self.clipnorm, self.decay, self.b1, self.b2, self.eps = lr, clipnorm, decay, beta1, beta2, eps
        self.iterations = 0
        self.m = OrderedDict((k, np.zeros_like(v)) for k, v in params.items())
        self.v = OrderedDict((k, np.zeros_like(v)) for k, v in params.items())

    def clip(self, grads):
        if self.clipnorm is None:
            return grads
        out = OrderedDict()
        for k, g in grads.items():
            n = np.sqrt(np.square(g.astype(np.float64)).sum())
            out[k] = g * (self.clipnorm / n) if n > self.clipnorm else g
        return out

    def apply(self, params, grads):
        dt = next(iter(params.values())).dtype.type
        grads = self.clip(grads)
        t = self.iterations + 1
        lr = dt(self.lr) / (dt(1.0) + dt(self.decay) * dt(self.iterations))
        lr_t = lr * np.sqrt(dt(1.0) - dt(self.b2) ** t) / (dt(1.0) - dt(self.b1) ** t)
        for k in params:
            g = grads[k]
            self.m[k] += (g - self.m[k]) * dt(1.0 - self.b1)
            self.v[k] += (g * g - self.v[k]) * dt(1.0 - self.b2)
            params[k] -= lr_t * self.m[k] / (np.sqrt(self.v[k]) + dt(self.eps))
        self.iterations += 1


class KerasImpalaOracle(object):
    """``ImpalaCnn.train`` / ``ImpalaMlp.train``: one ``model.fit`` epoch in minibatches of 128 (injected order)."""

    def __init__(self, spec, params, lr, ent_coef, clipnorm=None, decay=0.0, dtype=np.float64):
        self.net = ActorCritic(spec, params, dtype)
        self.ent_coef, self.dtype = ent_coef, dtype
        self.opt = AdamKeras(self.net.params, lr, clipnorm, decay)

    def predict(self, obs):
        logits, value = self.net.forward(obs)
        z = logits - logits.max(axis=1, keepdims=True)
        e = np.exp(z)
        return e / e.sum(axis=1, keepdims=True), value

    def step(self, obs, adv, onehot, target_v, apply=True):
        logits, value = self.net.forward(obs)
        loss, dlogits, dvalue, parts = keras_impala_loss_and_grads(logits, value, adv, onehot, target_v, self.ent_coef)
        grads = self.net.backward(dlogits, dvalue)
        if apply:
            self.opt.apply(self.net.params, grads)
        return dict(loss=loss, grads=grads, parts=parts, logits=logits, value=value, dlogits=dlogits, dvalue=dvalue)

    def fit(self, obs, adv, onehot, target_v, order, batch_size=128):
        """-> sample-weighted mean loss of the epoch (what Keras' History reports)."""
        tot = 0.0
        for lo in range(0, len(order), batch_size):
            idx = order[lo:lo + batch_size]
            tot += float(self.step(obs[idx], adv[idx], onehot[idx], target_v[idx])["loss"]) * len(idx)
        return tot / len(order)


class PpoLearnerOracle(object):
    """``PPO.train`` of xt/model/ppo/ppo.py:111-132 with injected permutations."""

    def __init__(self, spec, params, cfg, dtype=np.float64):
        self.net = ActorCritic(spec, params, dtype)
        self.cfg = cfg
        self.opt = AdamTF(self.net.params, cfg["LR"])
        self.dtype = dtype

    def step(self, obs, action, old_logp, adv, old_v, target_v, apply=True):
        c, dt = self.cfg, self.dtype
        logits, value = self.net.forward(obs)
        if self.net.spec.get("action_type") == "DiagGaussian":
            loss, dlogits, dvalue, dls, parts = gauss_ppo_loss_and_grads(
                logits, self.net.params["pi_logstd"], value, action, np.asarray(old_logp, dt), np.asarray(adv, dt),
                np.asarray(old_v, dt), np.asarray(target_v, dt),
                c["LOSS_CLIPPING"], c["ENTROPY_LOSS"], c["VF_CLIP"], c["CRITIC_LOSS_COEF"])
            grads = self.net.backward(dlogits, dvalue, extra={"pi_logstd": dls})
        else:
            loss, dlogits, dvalue, parts = ppo_loss_and_grads(
                logits, value, action, np.asarray(old_logp, dt), np.asarray(adv, dt),
                np.asarray(old_v, dt), np.asarray(target_v, dt),
                c["LOSS_CLIPPING"], c["ENTROPY_LOSS"], c["VF_CLIP"], c["CRITIC_LOSS_COEF"])
            grads = self.net.backward(dlogits, dvalue)
        clipped, gnorm = clip_by_global_norm(grads, c["MAX_GRAD_NORM"])
        if apply:
            self.opt.apply(self.net.params, clipped)
        return dict(loss=loss, logits=logits, value=value, dlogits=dlogits, dvalue=dvalue,
                    grads=grads, clipped=clipped, gnorm=gnorm, parts=parts)

    def train(self, state, label, perms):
        """perms: list (one per epoch) of index permutations replacing np.random.shuffle (:118)."""
        obs = state[0]
        nbatch = obs.shape[0]
        bs = self.cfg["BATCH_SIZE"]
        losses = []
        for ep in range(self.cfg["NUM_SGD_ITER"]):
            inds = np.asarray(perms[ep])
            for start in range(0, nbatch, bs):
                mb = inds[start:start + bs]
                # fed through float32 placeholders (xt/model/ppo/ppo.py:65-68)
                out = self.step(obs[mb], label[0][mb], label[1][mb].astype(np.float32),
                                label[2][mb].astype(np.float32), label[3][mb].astype(np.float32),
                                label[4][mb].astype(np.float32))
                losses.append(out["loss"])
        return np.mean(losses)


class ImpalaLearnerOracle(object):
    """``ImpalaCnnOpt.train`` (impala_cnn_opt.py:251-265) + ``IMPALAOpt.train`` chunking."""

    def __init__(self, spec, params, cfg, dtype=np.float64):
        self.net = ActorCritic(spec, params, dtype)
        self.cfg = cfg
        self.opt = RmsPropTF(self.net.params, cfg["LR"]) if cfg.get("opt_type") == "rmsprop" \
            else AdamTF(self.net.params, cfg["LR"])
        self.dtype = dtype

    def step(self, state, bp_logits, actions, dones, rewards, apply=True):
        c = self.cfg
        logits, value = self.net.forward(state)
        baseline = value[:, 0]
        loss, dlogits, dbase, parts = impala_loss_and_grads(
            logits, baseline, bp_logits, actions, dones, rewards, c["sample_batch_step"],
            c.get("GAMMA", 0.99), self.dtype)
        grads = self.net.backward(dlogits, dbase[:, None])
        clipped, gnorm = clip_by_global_norm(grads, c["grad_norm_clip"])
        if apply:
            self.opt.apply(self.net.params, clipped)
        return dict(loss=loss, logits=logits, baseline=baseline, dlogits=dlogits, dbaseline=dbase,
                    grads=grads, clipped=clipped, gnorm=gnorm, parts=parts)

    def train(self, states, bp_logits, actions, dones, rewards):
        """xt/algorithm/impala/impala_opt.py:73-106: sequential BATCH_SIZE chunks, mean of losses."""
        bs = self.cfg["BATCH_SIZE"]
        n = len(states)
        losses = []
        for s in range(0, n, bs):
            out = self.step(states[s:s + bs], bp_logits[s:s + bs], actions[s:s + bs],
                            dones[s:s + bs], np.asarray(rewards[s:s + bs], np.float32))
            losses.append(out["loss"])
        return np.mean(losses)
