"""Oracle (test infrastructure): a minimal eager ``tf`` look-alike backed by torch float64.

Purpose: EXECUTE the reference's own TensorFlow-side formula sources, unmodified, without
TensorFlow (not installed, not installable here):

    xt/model/ppo/__init__.py:4-25        actor_loss_with_entropy, critic_loss
    xt/model/tf_dist.py:49-130           DiagGaussianDist, CategoricalDist
    xt/model/impala/vtrace.py:39-115     from_logic_outputs
    xt/model/impala/impala_cnn_opt.py:171-196, 299-351   split_batches + loss wiring, vtrace_loss & co

``oracle/gen_golden_tf.py`` loads those files with this object installed as
``xt.model.tf_compat.tf`` and commits what they compute (values + autograd gradients) as
``tests/golden/tf_*.npz``.  Only the *primitive* ops below are restated; each one states the
TensorFlow 1.15 definition it follows (op name / gradient function in
tensorflow/python/ops/math_grad.py, nn_grad.py, clip_ops.py).  Every ``tf.float32`` is evaluated
in float64, so a fixture holds the exact-arithmetic value of the reference's formula (the fp32
kernels are compared with a tolerance, the float64 oracle restatement to ~1e-12).

Never imported by the product package.
"""
import contextlib
import types

import torch

F64 = torch.float64


class _Shape(tuple):
    """``Tensor.shape`` as TensorFlow's TensorShape where the reference uses it
    (vtrace.py:67-69 ``shape.assert_has_rank``; tf_dist.py:24 ``shape.as_list``)."""

    def assert_has_rank(self, rank):
        if len(self) != rank:
            raise ValueError("Shape %s must have rank %d" % (tuple(self), rank))

    def as_list(self):
        return list(self)


class T(torch.Tensor):
    """torch tensor whose ``.shape`` answers like a TensorShape; every op result stays a ``T``."""

    @property
    def shape(self):  # noqa: D401
        return _Shape(torch.Tensor.shape.__get__(self))


def _t(x, dtype=None):
    if isinstance(x, torch.Tensor):
        out = x if dtype is None or x.dtype == dtype else x.to(dtype)
    else:
        import numpy as np
        a = np.asarray(x)
        if dtype is None:
            dtype = F64 if a.dtype.kind == "f" else (torch.bool if a.dtype.kind == "b" else torch.int64)
        out = torch.as_tensor(a).to(dtype)
    return out if isinstance(out, T) else out.as_subclass(T)


class _MinimumTF(torch.autograd.Function):
    """tf.minimum; gradient = math_grad._MinimumGrad -> _MaximumMinimumGrad(op, grad, less_equal):
    ``xmask = x <= y``; dx = where(xmask, g, 0), dy = where(xmask, 0, g) (ties go to x)."""

    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x <= y)
        return torch.minimum(x, y)

    @staticmethod
    def backward(ctx, g):
        (xmask,) = ctx.saved_tensors
        z = torch.zeros_like(g)
        return torch.where(xmask, g, z), torch.where(xmask, z, g)


class _MaximumTF(torch.autograd.Function):
    """tf.maximum; gradient = math_grad._MaximumGrad: ``xmask = x >= y`` (ties go to x)."""

    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x >= y)
        return torch.maximum(x, y)

    @staticmethod
    def backward(ctx, g):
        (xmask,) = ctx.saved_tensors
        z = torch.zeros_like(g)
        return torch.where(xmask, g, z), torch.where(xmask, z, g)


def _bcast_pair(x, y):
    x, y = _t(x), _t(y)
    if x.dtype != y.dtype:
        y = y.to(x.dtype) if x.dtype.is_floating_point else y
        x = x.to(y.dtype) if y.dtype.is_floating_point and not x.dtype.is_floating_point else x
    return torch.broadcast_tensors(x, y)


def _unbroadcast(fn):
    # the autograd Functions above need equal shapes; broadcasting is done (differentiably) outside
    def op(x, y, name=None):
        xb, yb = _bcast_pair(x, y)
        return fn(xb.contiguous(), yb.contiguous())
    return op


minimum = _unbroadcast(_MinimumTF.apply)
maximum = _unbroadcast(_MaximumTF.apply)


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    """clip_ops.clip_by_value (TF 1.15): ``t_min = minimum(values, clip_value_max)``;
    ``maximum(t_min, clip_value_min)`` -> the gradient passes on the CLOSED interval."""
    t = _t(t)
    t = t if t.dtype.is_floating_point else t.to(F64)
    return maximum(minimum(t, float(clip_value_max)), float(clip_value_min))


def _axis(axis):
    return None if axis is None else (tuple(axis) if isinstance(axis, (list, tuple)) else int(axis))


def reduce_mean(x, axis=None, keepdims=False, name=None):
    x = _t(x)
    return x.mean() if axis is None else x.mean(dim=_axis(axis), keepdim=keepdims)


def reduce_sum(x, axis=None, keepdims=False, name=None):
    x = _t(x)
    return x.sum() if axis is None else x.sum(dim=_axis(axis), keepdim=keepdims)


def reduce_max(x, axis=None, keepdims=False, name=None):
    x = _t(x)
    return x.max() if axis is None else x.amax(dim=_axis(axis), keepdim=keepdims)


def one_hot(indices, depth, dtype=None):
    return torch.nn.functional.one_hot(_t(indices).long(), int(depth)).to(F64).as_subclass(T)


def _softmax_ce_dense(labels=None, logits=None, axis=-1, name=None):
    """tf.nn.softmax_cross_entropy_with_logits_v2: -sum(labels * log_softmax(logits)) along the class axis."""
    return -(_t(labels) * torch.log_softmax(_t(logits), dim=axis)).sum(dim=axis)


def _softmax_ce_sparse(labels=None, logits=None, name=None):
    """tf.nn.sparse_softmax_cross_entropy_with_logits: -log_softmax(logits)[label]."""
    logits = _t(logits)
    lab = _t(labels).long()
    return -torch.gather(torch.log_softmax(logits, dim=-1), -1, lab.unsqueeze(-1)).squeeze(-1)


def scan(fn, elems, initializer=None, parallel_iterations=10, back_prop=True, swap_memory=False,
         infer_shape=True, reverse=False, name=None):
    """tf.scan over the leading axis of a tuple of tensors (functional_ops.scan); ``reverse=True`` walks
    from the last element to the first and returns the results in the original order."""
    single = not isinstance(elems, (tuple, list))
    seq = (elems,) if single else tuple(elems)
    n = seq[0].shape[0]
    order = range(n - 1, -1, -1) if reverse else range(n)
    acc = initializer
    out = [None] * n
    for i in order:
        item = seq[0][i] if single else tuple(s[i] for s in seq)
        acc = fn(acc, item)
        out[i] = acc
    res = torch.stack(out, dim=0)
    return res if back_prop else res.detach()


def concat(values, axis=0, name=None):
    parts = []
    for v in values:
        if isinstance(v, (list, tuple)):
            v = torch.stack([_t(e).reshape(()) for e in v])
        parts.append(_t(v))
    if not any(p.dtype.is_floating_point for p in parts):
        parts = [p.long() for p in parts]
    return torch.cat(parts, dim=int(axis))


def reshape(x, shape, name=None):
    if isinstance(shape, torch.Tensor):
        shape = [int(s) for s in shape.tolist()]
    return _t(x).reshape([int(s) for s in shape])


def split(value, num_or_size_splits, axis=0, name=None):
    value = _t(value)
    if isinstance(num_or_size_splits, int):
        return list(torch.chunk(value, num_or_size_splits, dim=axis))
    return list(torch.split(value, list(num_or_size_splits), dim=axis))


@contextlib.contextmanager
def _nullctx(*a, **k):
    yield


def make_tf():
    """The object installed as ``xt.model.tf_compat.tf``."""
    tf = types.SimpleNamespace()
    tf.float32, tf.float64, tf.int32, tf.int64, tf.bool = F64, F64, torch.int64, torch.int64, torch.bool
    tf.exp = lambda x, name=None: torch.exp(_t(x))
    tf.log = lambda x, name=None: torch.log(_t(x))
    tf.square = lambda x, name=None: _t(x) * _t(x)
    tf.add = lambda x, y, name=None: _t(x) + _t(y)
    tf.minimum, tf.maximum, tf.clip_by_value = minimum, maximum, clip_by_value
    tf.reduce_mean, tf.reduce_sum, tf.reduce_max = reduce_mean, reduce_sum, reduce_max
    tf.one_hot = one_hot
    tf.expand_dims = lambda x, axis=None, name=None: _t(x).unsqueeze(int(axis))
    tf.squeeze = lambda x, axis=None, name=None: _t(x).squeeze() if axis is None else _t(x).squeeze(_axis(axis))
    tf.cast = lambda x, dtype, name=None: _t(x).to(dtype)
    tf.convert_to_tensor = lambda x, dtype=None, name=None: _t(x, dtype)
    tf.shape = lambda x, name=None: torch.tensor(list(_t(x).shape), dtype=torch.int64).as_subclass(T)
    tf.zeros_like = lambda x, name=None: torch.zeros_like(_t(x))
    tf.stop_gradient = lambda x, name=None: _t(x).detach()
    tf.concat, tf.reshape, tf.split, tf.scan = concat, reshape, split, scan
    tf.transpose = lambda x, perm=None, name=None: _t(x).permute([int(p) for p in perm])
    tf.device = _nullctx
    tf.variable_scope = _nullctx
    tf.nn = types.SimpleNamespace(
        softmax_cross_entropy_with_logits_v2=_softmax_ce_dense,
        sparse_softmax_cross_entropy_with_logits=_softmax_ce_sparse,
        softmax=lambda x, axis=-1, name=None: torch.softmax(_t(x), dim=axis),
        log_softmax=lambda x, axis=-1, name=None: torch.log_softmax(_t(x), dim=axis),
    )
    tf.__version__ = "1.15.0"
    return tf
