"""Inference-only CPU replica of the actor-critic network.

XingTian constructs the configured ``Model`` class not only in the learner but also in every explorer, evaluator and
predictor process, with ``CUDA_VISIBLE_DEVICES=-1`` (xt/framework/explorer.py:60, predictor.py:88) -- there the model
only ever runs ``predict`` (xt/agent/ppo/ppo.py:43; xt/model/ppo/ppo.py:104-109; impala_cnn_opt.py:267-277) and
receives weights by TF variable name (xt/framework/agent_group.py:285 -> ``set_weights``).  This module is that
replica: the same ``NetSpec`` (layers, flat fp32 parameter layout, TF variable names) evaluated with numpy -- NHWC
im2col + one sgemm per layer -- so that weights published by the GPU learner drop straight in.

It is NOT a fallback of the learner update: it has no ``train``; ``model_info["type"] == "learner"``
(xt/framework/learner.py:544) always builds the HIP network and fails loudly without a GPU.
"""
from collections import OrderedDict

import numpy as np


def initial_weights(spec, seed=None, baseline_norm_std=None):
    """Keras default initialisation of every variable of ``spec`` (glorot_uniform kernels, zero biases; Conv2D/Dense
    in xt/model/model_utils.py:86-96) and, for ImpalaCnnOpt's baseline, ``custom_norm_initializer(std)``
    (xt/model/model_utils.py:204-211).  One generator, fixed variable order -> the GPU learner and a CPU replica
    built from the same seed start identical."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for name, (_, shape) in spec.names.items():
        if name.endswith("/bias") or name == "pi_logstd":
            w[name] = np.zeros(shape, np.float32)
        else:
            if len(shape) == 4:
                rf = shape[0] * shape[1]
                fan_in, fan_out = rf * shape[2], rf * shape[3]
            else:
                fan_in, fan_out = shape
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            w[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
    if baseline_norm_std is not None:
        shape = spec.names[spec.v_name + "/kernel"][1]
        out = rng.standard_normal(shape).astype(np.float32)
        out *= baseline_norm_std / np.sqrt(np.square(out).sum(axis=0, keepdims=True))
        w[spec.v_name + "/kernel"] = out
    return w


def _act(z, act):
    """ACTIVATION_MAP of xt/model/model_utils.py:8-20 (the monotonic entries the HIP learner trains with)."""
    if act == "relu":
        return np.maximum(z, 0.0, out=z)
    if act == "tanh":
        return np.tanh(z, out=z)
    if act == "sigmoid":
        return (1.0 / (1.0 + np.exp(-z))).astype(z.dtype)
    if act == "softsign":
        return z / (1.0 + np.abs(z))
    if act == "softplus":
        return np.logaddexp(0.0, z).astype(z.dtype)
    if act == "leaky_relu":
        return np.where(z > 0, z, np.float32(0.2) * z)
    if act == "elu":
        return np.where(z > 0, z, np.expm1(np.minimum(z, 0)))
    if act == "selu":
        return (1.0507009873554805 * np.where(z > 0, z, 1.6732632423543772 * np.expm1(np.minimum(z, 0)))).astype(z.dtype)
    if act == "swish":
        return (z / (1.0 + np.exp(-z))).astype(z.dtype)
    if act == "gelu":                    # xt/model/tf_utils.py:157-166
        return (0.5 * z * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (z + 0.044715 * z ** 3)))).astype(z.dtype)
    if act in (None, "none"):
        return z
    raise KeyError("activation {} not implemented.".format(act))


class CpuActorCritic(object):
    """``forward`` / ``get_weights`` / ``set_weights`` of ``HipActorCritic`` on the host, float32."""

    inference_only = True

    def __init__(self, spec, max_batch=None, seed=None, init="glorot"):
        self.spec = spec
        self.params = np.zeros(spec.n_flat, np.float32)
        if init == "glorot":
            self.init_weights(seed)

    def init_weights(self, seed=None, baseline_norm_std=None):
        self.set_weights(initial_weights(self.spec, seed, baseline_norm_std))

    # ------------------------------------------------------------------ weights by TF variable name
    def get_weights(self):
        out = OrderedDict()
        for name in self.spec.names:
            out[name] = self.spec.var_view(self.params, name).copy()
        return out

    def set_weights(self, weights):
        """Unknown names are ignored, KeyError if nothing matches (TFVariables.set_weights, tf_utils.py:104-128)."""
        hit = [k for k in weights.keys() if k in self.spec.names]
        if not hit:
            raise KeyError("NO node's weights could assign in self.graph {} vs {}".format(
                list(self.spec.names.keys()), list(weights.keys())))
        for name in hit:
            off, shape = self.spec.names[name]
            val = np.asarray(weights[name], np.float32)
            if tuple(val.shape) != tuple(shape):
                raise KeyError("update {} encounter error: shape {} vs {}".format(name, val.shape, shape))
            self.spec.var_view(self.params, name)[...] = val

    def publish_weights(self, ring, ctr_info=None, lag=0):
        """hand the replica's weights to a ``transport.WeightsRing`` in the packed form the learner publishes"""
        return ring.publish_flat_host(self.params, self.spec, ctr_info)

    # ------------------------------------------------------------------ forward
    def _layer(self, lay, x):
        """x: [B, H, W, C] float32 -> [B, OH, OW, N]; Conv2D (VALID / TensorFlow's asymmetric SAME) or Dense."""
        p = self.params
        w = p[lay.param_off:lay.param_off + lay.K * lay.N].reshape(lay.K, lay.N)
        b = p[lay.param_off + lay.K * lay.N:lay.param_off + (lay.K + 1) * lay.N]
        bsz = x.shape[0]
        if lay.KH == 1 and lay.KW == 1 and lay.S == 1:
            cols = x.reshape(bsz * lay.OH * lay.OW, lay.C)
        else:
            pb = max((lay.OH - 1) * lay.S + lay.KH - lay.H - lay.PT, 0)
            pr = max((lay.OW - 1) * lay.S + lay.KW - lay.W - lay.PL, 0)
            if lay.PT or lay.PL or pb or pr:
                x = np.pad(x, ((0, 0), (lay.PT, pb), (lay.PL, pr), (0, 0)))
            win = np.lib.stride_tricks.sliding_window_view(x, (lay.KH, lay.KW), axis=(1, 2))   # [B,H',W',C,KH,KW]
            win = win[:, ::lay.S, ::lay.S][:, :lay.OH, :lay.OW]
            cols = np.ascontiguousarray(win.transpose(0, 1, 2, 4, 5, 3)).reshape(bsz * lay.OH * lay.OW, lay.K)
        z = cols @ w
        z += b
        return _act(z, lay.act).reshape(bsz, lay.OH, lay.OW, lay.N)

    def forward(self, obs):
        """obs [B, ...] -> (logits [B, A] (the mean for DiagGaussian), value [B]) as numpy float32."""
        spec = self.spec
        x0 = np.asarray(obs)
        is_u8, mean, std = spec.input_xform
        lay0 = spec.layers[0]
        if is_u8:
            x0 = x0.astype(np.float32)
            x0 = (x0 - np.float32(mean)) / np.float32(std) if mean != 0.0 else x0 / np.float32(std)
        else:
            x0 = x0.astype(np.float32)
        if x0.ndim == 2:                                     # MLP observations, zero-padded to a multiple of 4
            if x0.shape[1] < lay0.C:
                x0 = np.pad(x0, ((0, 0), (0, lay0.C - x0.shape[1])))
            x0 = x0.reshape(x0.shape[0], 1, 1, lay0.C)
        elif x0.ndim == 4 and x0.shape[3] < lay0.C:          # image channels zero-padded to a multiple of 4 (netspec._conv)
            x0 = np.pad(x0, ((0, 0), (0, 0), (0, 0), (0, lay0.C - x0.shape[3])))
        feats = []
        for tr in range(spec.n_trunks):
            x = x0
            for lay in spec.layers:
                if lay.trunk != tr:
                    continue
                if lay.H == 1 and lay.W == 1 and x.shape[1:3] != (1, 1):
                    x = x.reshape(x.shape[0], 1, 1, -1)      # Flatten in (H, W, C) order
                x = self._layer(lay, x)
            feats.append(x.reshape(x.shape[0], -1))
        f_pi, f_v = feats[0], feats[-1]
        p, f, a = self.params, spec.feat, spec.action_dim
        wpi = p[spec.pi_off:spec.pi_off + f * a].reshape(f, a)
        bpi = p[spec.pi_off + f * a:spec.pi_off + f * a + a]
        wv = p[spec.v_off:spec.v_off + f]
        bv = p[spec.v_off + f]
        return f_pi @ wpi + bpi, f_v @ wv + bv
