"""Oracle (test infrastructure): return computations of the hot path in numpy.

* ``gae``            restates ``xt/agent/ppo/ppo.py:77-106`` (``PPO.data_proc``)
* ``vtrace_from_logits`` restates ``xt/model/impala/vtrace.py:39-115``
  (``from_logic_outputs``), which is TensorFlow graph code in the reference.
* ``split_batches``  restates ``xt/model/impala/impala_cnn_opt.py:171-186``.
"""
import numpy as np

GAMMA = 0.99  # xt/agent/ppo/default_config.py:2, xt/model/impala/default_config.py:5
LAM = 0.95    # xt/agent/ppo/default_config.py:3


def gae(value, reward, done, gamma=GAMMA, lam=LAM):
    """GAE(gamma, lambda) exactly as the reference actor computes it.

    value  : [T+1, 1] float32  (V(s_0..s_T); last row is the bootstrap value,
             appended in ``get_trajectory`` xt/agent/ppo/ppo.py:71-75)
    reward : [T] float64 (python floats in the reference)
    done   : [T] bool
    returns (adv [T,1] f64, old_value [T,1] f32, target_value [T,1] f64)

    Same numpy expression order as xt/agent/ppo/ppo.py:87-104 so that the result
    is bit-identical to the reference: ``discount = ~done * GAMMA`` (:92),
    ``delta = reward + discount*next_value - value`` (:93), the in-place reverse
    loop ``adv[j] += adv[j+1] * discount[j] * LAM`` (:96-97) and
    ``target_value = adv + value`` (:104).
    """
    value = np.asarray(value)
    reward = np.asarray(reward)
    done = np.asarray(done)
    next_value = value[1:]
    value = value[:-1]
    done = np.expand_dims(done, axis=1)
    reward = np.expand_dims(reward, axis=1)
    discount = ~done * gamma
    delta_t = reward + discount * next_value - value
    adv = delta_t
    for j in range(len(adv) - 2, -1, -1):
        adv[j] += adv[j + 1] * discount[j] * lam
    return adv, value, adv + value


def split_batches(x, batch_step, drop_last=False):
    """[B*T, ...] (env-major, index b*T+t) -> [T, B, ...]; impala_cnn_opt.py:171-186."""
    x = np.asarray(x)
    batch_count = x.shape[0] // batch_step
    res = x.reshape((batch_count, batch_step) + x.shape[1:])
    res = np.swapaxes(res, 0, 1)
    if drop_last:
        return res[:-1]
    return res


def _log_softmax(logits):
    m = logits.max(axis=-1, keepdims=True)
    z = logits - m
    return z - np.log(np.exp(z).sum(axis=-1, keepdims=True))


def sparse_softmax_ce(logits, labels):
    """tf.nn.sparse_softmax_cross_entropy_with_logits: -log_softmax(logits)[label]."""
    lsm = _log_softmax(logits)
    return -np.take_along_axis(lsm, labels[..., None].astype(np.int64), axis=-1)[..., 0]


def vtrace_from_logits(bp_logits, tp_logits, actions, discounts, rewards, values,
                       bootstrap_value, clip_rho=1.0, clip_pg_rho=1.0, dtype=np.float64):
    """V-trace targets from behaviour/target logits; time-major [T', B(, A)].

    Follows xt/model/impala/vtrace.py: log-rhos from two sparse softmax-CEs
    (:71-78), rho_bar/c clipping (:80-85), deltas (:90), the reverse scan
    ``acc = delta_t + discount_t * c_t * acc`` (:94-106), ``vs = acc + V`` (:108)
    and ``pg_adv = rho_pg * (r + discount * vs_{t+1} - V)`` (:110-111).
    """
    bp_logits = np.asarray(bp_logits, dtype)
    tp_logits = np.asarray(tp_logits, dtype)
    discounts = np.asarray(discounts, dtype)
    rewards = np.asarray(rewards, dtype)
    values = np.asarray(values, dtype)
    bootstrap_value = np.asarray(bootstrap_value, dtype)
    target_log_prob = -sparse_softmax_ce(tp_logits, actions)
    behaviour_log_prob = -sparse_softmax_ce(bp_logits, actions)
    rhos = np.exp(target_log_prob - behaviour_log_prob)
    clipped_rhos = np.minimum(dtype(clip_rho), rhos)
    clipped_pg_rhos = np.minimum(dtype(clip_pg_rho), rhos)
    cs = np.minimum(dtype(1.0), rhos)
    next_values = np.concatenate([values[1:], bootstrap_value[None]], axis=0)
    deltas = clipped_rhos * (rewards + discounts * next_values - values)
    acc = np.zeros_like(bootstrap_value)
    out = np.zeros_like(values)
    for t in range(values.shape[0] - 1, -1, -1):
        acc = deltas[t] + discounts[t] * cs[t] * acc
        out[t] = acc
    vs = out + values
    vs_next = np.concatenate([vs[1:], bootstrap_value[None]], axis=0)
    pg_adv = clipped_pg_rhos * (rewards + discounts * vs_next - values)
    return vs, pg_adv
