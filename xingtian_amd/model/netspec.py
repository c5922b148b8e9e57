"""Network descriptions (layer geometry + flat parameter layout) for the HIP learner.

Restates the architecture tables of the reference: ``get_default_filters``
(xt/model/model_utils.py:120-149), ``get_cnn_backbone`` / ``get_mlp_backbone``
(:22-80), ``get_atari_filter`` (xt/model/atari_model.py:4-23) and
``ImpalaCnnOpt.create_model`` (xt/model/impala/impala_cnn_opt.py:110-152), with
TensorFlow's VALID/SAME output-size and (asymmetric) padding rules.

Flat fp32 parameter buffer layout (ours, not TF's creation order): every trunk layer's
``[K*N] kernel + [N] bias`` block, then the pi head ``[F*A]+[A]``, then the v head
``[F]+[1]``; every block starts on a 16-byte boundary.  ``names`` maps the reference's TF
variable names (Keras layer names, model_utils.py:34-36,87,96) onto (offset, shape).
"""
from collections import OrderedDict


def conv_out(size, k, s, padding):
    if padding == "valid":
        return (size - k) // s + 1, 0
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2          # TF SAME: the odd cell goes after (bottom/right)


def ppo_cnn_filters(state_dim):
    hw = list(state_dim[:2])
    if len(state_dim) != 3:
        raise ValueError("Without default architecture for obs shape {}".format(state_dim))
    if hw == [84, 84]:
        return [(32, 8, 4), (32, 4, 2), (64, 3, 1)]
    if hw == [42, 42]:
        return [(32, 4, 2), (32, 4, 2), (64, 3, 1)]
    if hw == [15, 15]:
        return [(32, 5, 1), (64, 3, 1), (64, 3, 1)]
    if hw[0] != hw[1]:
        raise ValueError("Without default architecture for non-square obs shape {}".format(state_dim))
    return infer_filters(hw[0])


def _infer_stride_and_kernel(size, flat):
    """xt/model/model_utils.py:165-176 -> (kernel, stride, flat)."""
    if flat or size <= 3:
        return 1, 1, True
    if size <= 8:
        return 3, 1, True
    if size <= 64:
        return 5, 2, False
    raise ValueError("Without default architecture for obs shape > 64 (the reference's rule yields a kernel larger "
                     "than the image there)")


def infer_filters(size):
    """The reference's fallback for square observations without a table (model_utils.py:150-162): 16, 32, 64, ...
    filters, 5x5/2 while the (floor-divided) size is above 8, then one 3x3/1 (or 1x1/1 at <= 3) layer."""
    filters, flat, n = [], False, 16
    while not flat:
        k, s, flat = _infer_stride_and_kernel(size, flat)
        filters.append((n, k, s))
        n *= 2
        size //= s
    return filters


def impala_filters(state_dim):
    hw = list(state_dim[:2])
    if len(state_dim) == 3 and hw == [84, 84]:
        return [(16, 8, 4), (32, 4, 2), (256, 11, 1)]
    if len(state_dim) == 3 and hw == [42, 42]:
        return [(16, 4, 2), (32, 4, 2), (256, 11, 1)]
    raise ValueError("Without default architecture for obs shape {}".format(state_dim))


class Layer(object):
    __slots__ = ("name", "H", "W", "C", "KH", "KW", "S", "PT", "PL", "OH", "OW", "N", "act", "trunk",
                 "param_off", "kernel_shape", "store_shape")

    @property
    def K(self):
        return self.KH * self.KW * self.C


def _conv(name, h, w, c, cout, k, s, padding, act, trunk):
    lay = Layer()
    # an image whose channel count is not a multiple of 4 (examples/ant_ppo.yaml: [84, 84, 3]; the reference's
    # get_cnn_backbone takes any C, xt/model/model_utils.py:49-80) enters the layer kernels with zero planes up to the
    # next multiple of 4 (xt_pad_channels): the TF kernel [k, k, c, cout] lives inside a [k, k, pad4(c), cout] block
    # whose extra input-channel rows start at 0 and stay 0 (zero operand -> zero gradient -> zero Adam step) -- exact.
    cpad = (c + 3) & ~3
    lay.name, lay.H, lay.W, lay.C, lay.KH, lay.KW, lay.S = name, h, w, cpad, k, k, s
    lay.OH, lay.PT = conv_out(h, k, s, padding)
    lay.OW, lay.PL = conv_out(w, k, s, padding)
    lay.N, lay.act, lay.trunk = cout, act, trunk
    lay.kernel_shape = (k, k, c, cout)
    lay.store_shape = (k, k, cpad, cout) if cpad != c else None
    if cpad != c and padding == "valid" and k == h == w:
        raise ValueError("layer {}: a whole-image kernel on {} channels is not supported".format(name, c))
    if padding == "valid" and k == h == w:
        # a VALID conv whose kernel covers the whole image (ImpalaCnnOpt's 11x11 -> 1x1x256,
        # impala_cnn_opt.py:129-136) IS a dense layer on the NHWC-flattened input: HWIO kernel memory
        # [ky,kx,c][n] == [in,out].  Run it as Dense so that the input gradient is one GEMM instead of a
        # 121-tap gather of which a single tap is valid per pixel.  The TF kernel shape is kept for naming.
        lay.H = lay.W = lay.KH = lay.KW = lay.S = 1
        lay.C = h * w * c
    return lay


def _dense(name, cin, cout, act, trunk):
    lay = Layer()
    lay.name, lay.H, lay.W, lay.C, lay.KH, lay.KW, lay.S = name, 1, 1, cin, 1, 1, 1
    lay.OH = lay.OW = 1
    lay.PT = lay.PL = 0
    lay.N, lay.act, lay.trunk = cout, act, trunk
    lay.kernel_shape = (cin, cout)
    lay.store_shape = None
    return lay


class NetSpec(object):
    """layers (trunk 0 first), heads, flat layout, name map."""

    def __init__(self, layers, n_trunks, feat, action_dim, pi_name, v_name, pi_kernel_shape, input_xform,
                 state_dim, action_type="Categorical"):
        if action_type not in ("Categorical", "DiagGaussian"):
            raise NotImplementedError(
                "action type: {} not match any implemented distributions.".format(action_type))
        self.action_type = action_type
        self.layers, self.n_trunks, self.feat, self.action_dim = layers, n_trunks, feat, action_dim
        self.pi_name, self.v_name, self.input_xform, self.state_dim = pi_name, v_name, input_xform, tuple(state_dim)
        off = 0
        self.names = OrderedDict()
        self.store_shape = {}       # name -> shape of the block in the flat buffer, where it differs from the TF shape
        for lay in layers:
            if lay.C % 4 or lay.N % 4:
                raise ValueError("layer {}: channel counts must be multiples of 4 for the HIP kernels "
                                 "(C={}, N={})".format(lay.name, lay.C, lay.N))
            lay.param_off = off
            self.names[lay.name + "/kernel"] = (off, lay.kernel_shape)
            if getattr(lay, "store_shape", None):
                self.store_shape[lay.name + "/kernel"] = tuple(lay.store_shape)
            self.names[lay.name + "/bias"] = (off + lay.K * lay.N, (lay.N,))
            off += (lay.K + 1) * lay.N
            off = (off + 3) & ~3
        self.pi_off = off
        self.names[pi_name + "/kernel"] = (off, pi_kernel_shape)
        self.names[pi_name + "/bias"] = (off + feat * action_dim, (action_dim,))
        off = (off + feat * action_dim + action_dim + 3) & ~3
        self.v_off = off
        self.names[v_name + "/kernel"] = (off, (feat, 1))
        self.names[v_name + "/bias"] = (off + feat, (1,))
        off = (off + feat + 1 + 3) & ~3
        self.logstd_off = 0
        if action_type == "DiagGaussian":       # tf.get_variable('pi_logstd', (1, A), zeros), xt/model/ppo/ppo.py:78
            self.logstd_off = off
            self.names["pi_logstd"] = (off, (1, action_dim))
            off = (off + action_dim + 3) & ~3
        self.n_flat = off
        self.n_params = sum(int(_prod(s)) for _, s in self.names.values())

    def var_view(self, flat, name):
        """The variable ``name`` (TF shape) as a VIEW into the flat numpy buffer ``flat`` -- contiguous for every block
        but a channel-padded first-layer kernel, whose TF tensor is a strided sub-block of its storage."""
        off, shape = self.names[name]
        st = self.store_shape.get(name)
        if st is None:
            return flat[off:off + int(_prod(shape))].reshape(shape)
        return flat[off:off + int(_prod(st))].reshape(st)[tuple(slice(0, d) for d in shape)]

    def var_extent(self, name):
        """(offset, number of floats) the variable occupies in the flat buffer (its storage block, padding included)."""
        off, shape = self.names[name]
        return off, int(_prod(self.store_shape.get(name, shape)))

    @property
    def obs_channels_padded(self):
        """channels of the observation as the first layer reads it (>= the reference's state_dim[-1]) for image
        inputs, else None"""
        lay0 = self.layers[0]
        if len(self.state_dim) == 3 and not (lay0.H == lay0.W == 1):
            return lay0.C if lay0.C != int(self.state_dim[2]) else None
        return None


def _prod(shape):
    p = 1
    for s in shape:
        p *= s
    return p


def _mlp(prefix, cin, hidden_sizes, act, trunk):
    out = []
    for i, hs in enumerate(hidden_sizes):
        lay = _dense("{}_hidden_mlp_{}".format(prefix, i), (cin + 3) & ~3, hs, act, trunk)
        # a feature count that is not a multiple of 4 (Pendulum: 3) is zero-padded: the observation gets zero
        # columns (HipActorCritic.to_device_obs) and the [cin,N] TF kernel is the head of a [pad4(cin),N] block
        # whose extra rows start at 0 and stay 0 (their gradient is sum(0 * dz) = 0, Adam of 0 is 0) -- exact.
        lay.kernel_shape = (cin, hs)
        out.append(lay)
        cin = hs
    return out, cin


def ppo_cnn(state_dim, action_dim, hidden_sizes=(512,), act="relu", vf_share=True, input_dtype="uint8",
            action_type="Categorical"):
    if input_dtype not in ("uint8", "float32"):      # get_cnn_backbone, xt/model/model_utils.py:53-58
        raise ValueError("dtype: {} not supported automatically, please implement it yourself".format(input_dtype))
    layers, feat = [], None
    for trunk, prefix in enumerate(["shared"] if vf_share else ["pi", "v"]):
        h, w, c = state_dim
        for i, (cout, k, s) in enumerate(ppo_cnn_filters(state_dim)):
            lay = _conv("{}_conv_layer_{}".format(prefix, i), h, w, c, cout, k, s, "valid", act, trunk)
            layers.append(lay)
            h, w, c = lay.OH, lay.OW, cout
        mlps, feat = _mlp(prefix, h * w * c, hidden_sizes, act, trunk)
        layers += mlps
    xf = (1, 0.0, 255.0) if input_dtype == "uint8" else (0, 0.0, 1.0)
    return NetSpec(layers, 1 if vf_share else 2, feat, action_dim, "pi_latent", "output_value",
                   (feat, action_dim), xf, state_dim, action_type)


def ppo_mlp(state_dim, action_dim, hidden_sizes=(64, 64), act="tanh", vf_share=False, action_type="Categorical"):
    layers, feat = [], None
    for trunk, prefix in enumerate(["shared"] if vf_share else ["pi", "v"]):
        mlps, feat = _mlp(prefix, int(state_dim[0]), hidden_sizes, act, trunk)
        layers += mlps
    return NetSpec(layers, 1 if vf_share else 2, feat, action_dim, "pi_latent", "output_value",
                   (feat, action_dim), (0, 0.0, 1.0), (1, 1, int(state_dim[0])), action_type)


def impala_cnn_opt(state_dim, action_dim, state_mean=0.0, state_std=255.0, input_dtype="uint8"):
    filt = impala_filters(state_dim)
    names = ["explore_agent/conv2d", "explore_agent/conv2d_1", "explore_agent/conv2d_2"]
    h, w, c = state_dim
    layers = []
    for i, (cout, k, s) in enumerate(filt):
        lay = _conv(names[i], h, w, c, cout, k, s, "same" if i < len(filt) - 1 else "valid", "relu", 0)
        layers.append(lay)
        h, w, c = lay.OH, lay.OW, cout
    if (h, w) != (1, 1):
        raise ValueError("ImpalaCnnOpt expects the last conv to collapse to 1x1, got {}x{}".format(h, w))
    if input_dtype in ("float32", "float", "float64"):
        xf = (0, 0.0, 1.0)
    else:
        xf = (1, float(state_mean), float(state_std))
    return NetSpec(layers, 1, c, action_dim, "explore_agent/conv2d_3", "explore_agent/dense",
                   (1, 1, c, action_dim), xf, state_dim)


def impala_cnn(state_dim, action_dim):
    """Keras ``ImpalaCnn`` (xt/model/impala/impala_cnn.py:44-57): Conv2D 32@8x8/4, 64@4x4/2, 64@3x3/1 (valid, relu) on
    uint8 / 255, Dense 256 relu; a softmax policy head and a value head on the same features.  Variable names are
    Keras' automatic layer names (conv2d, conv2d_1, conv2d_2, dense) and the two named output layers."""
    h, w, c = state_dim
    layers = []
    for name, (cout, k, s) in zip(("conv2d", "conv2d_1", "conv2d_2"), ((32, 8, 4), (64, 4, 2), (64, 3, 1))):
        lay = _conv(name, h, w, c, cout, k, s, "valid", "relu", 0)
        layers.append(lay)
        h, w, c = lay.OH, lay.OW, cout
    layers.append(_dense("dense", h * w * c, 256, "relu", 0))
    return NetSpec(layers, 1, 256, action_dim, "output_actions", "output_value", (256, action_dim), (1, 0.0, 255.0),
                   state_dim)


def impala_mlp(state_dim, action_dim, hidden_size=128, num_layers=1):
    """Keras ``ImpalaMlp`` (xt/model/impala/impala_mlp.py:42-53): NUM_LAYERS x Dense(HIDDEN_SIZE, relu) named dense,
    dense_1, ...; softmax policy + value heads.  An input width that is not a multiple of 4 is zero-padded (see _mlp)."""
    layers, cin = [], int(state_dim[0])
    for i in range(int(num_layers)):
        lay = _dense("dense" if i == 0 else "dense_%d" % i, (cin + 3) & ~3, hidden_size, "relu", 0)
        lay.kernel_shape = (cin, hidden_size)
        layers.append(lay)
        cin = hidden_size
    return NetSpec(layers, 1, hidden_size, action_dim, "output_actions", "output_value", (hidden_size, action_dim),
                   (0, 0.0, 1.0), (1, 1, int(state_dim[0])))
