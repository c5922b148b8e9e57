"""Oracle (test infrastructure): golden vectors of the reference's ALGORITHM-level host logic, produced by
EXECUTING the reference.

``xt/algorithm/ppo/ppo.py::PPO``, ``xt/algorithm/impala/impala_opt.py::IMPALAOpt`` and
``xt/algorithm/impala/impala.py::IMPALA`` (with their real base class
``xt/algorithm/algorithm.py``, ``xt/algorithm/alg_utils.py``, ``zeus/common/util/common.py::import_config`` and
``xt/algorithm/impala/default_config.py``) are loaded from /root/reference with importlib.  Only what cannot be
imported here is stubbed: ``absl.logging`` (-> stdlib logging), ``zeus.common.util.register.Registers``
(decorators that return the class), ``zeus.common.ipc.uni_comm.UniComm`` (never used: use_train_thread is False),
``xt.model.tf_compat.loss_to_val`` (identity for plain floats, as in the reference for non-Keras losses) and
``xt.model.model_builder``, which returns a RECORDING model: every ``train(state, label)`` call is stored, so the
fixture pins what the framework hands to ``Model.train`` -- concatenation order of ragged trajectories, dtypes,
IMPALA's sequential BATCH_SIZE chunking, the loss it returns.  One numpy alias is restored (``np.bool``, removed in
numpy 1.24, used by impala_opt.py:144).

Writes tests/golden/alg_ppo.npz, alg_impala_opt.npz and alg_impala.npz (the non-opt ``IMPALA``: actor forward over all
stored states, numpy v-trace on probabilities, BATCH_SIZE chunks); nothing at test time reads /root/reference.

Usage:  python oracle/gen_golden_alg.py
"""
import importlib.util
import logging as pylogging
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class RecordingModel(object):
    """Stands in for the TF model: remembers what Model.train receives and returns 1.5, 2.5, ... as 'losses'."""

    def __init__(self, model_info):
        self.model_info = model_info
        self.calls = []

    def train(self, state, label):
        cp = lambda a: [np.array(x, copy=True) for x in a] if isinstance(a, (list, tuple)) else np.array(a, copy=True)
        self.calls.append((cp(state), cp(label)))
        return 0.5 + len(self.calls)

    def predict(self, state):
        """IMPALA (non-opt) runs the actor over every stored state before its numpy v-trace: deterministic
        probabilities / values, kept so that the test's stand-in model can return the very same arrays."""
        n, a_dim = len(state[0]), self.model_info["action_dim"]
        rng = np.random.default_rng(4242 + n)
        logits = rng.standard_normal((n, a_dim))
        p = np.exp(logits - logits.max(-1, keepdims=True))
        self.pred = [(p / p.sum(-1, keepdims=True)).astype(np.float32), rng.standard_normal((n, 1)).astype(np.float32)]
        self.pred_state = np.array(state[0], copy=True)
        return self.pred


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_algorithms():
    if not hasattr(np, "bool"):
        np.bool = bool                       # numpy < 1.24 alias used by the reference

    class _Stub(object):
        def __call__(self, cls):
            return cls

    class _Registers(object):
        algorithm = _Stub()

    def pkg(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    absl = pkg("absl")
    absl.logging = pylogging
    sys.modules["absl.logging"] = pylogging
    for n in ("xt", "xt.model", "xt.algorithm", "xt.algorithm.impala", "xt.algorithm.ppo", "zeus", "zeus.common",
              "zeus.common.util", "zeus.common.ipc"):
        pkg(n)
    sys.modules["xt.model"].model_builder = lambda model_info: RecordingModel(model_info)
    tfc = types.ModuleType("xt.model.tf_compat")
    tfc.loss_to_val = lambda loss: loss
    sys.modules["xt.model.tf_compat"] = tfc
    reg = types.ModuleType("zeus.common.util.register")
    reg.Registers = _Registers
    sys.modules["zeus.common.util.register"] = reg
    uni = types.ModuleType("zeus.common.ipc.uni_comm")
    uni.UniComm = object
    sys.modules["zeus.common.ipc.uni_comm"] = uni
    _load("zeus.common.util.common", "zeus/common/util/common.py")            # real import_config
    _load("xt.algorithm.alg_utils", "xt/algorithm/alg_utils.py")               # real dist policies
    base = _load("xt.algorithm.algorithm", "xt/algorithm/algorithm.py")        # real Algorithm base class
    sys.modules["xt.algorithm"].Algorithm = base.Algorithm
    _load("xt.algorithm.impala.default_config", "xt/algorithm/impala/default_config.py")
    ppo = _load("xt.algorithm.ppo.ppo", "xt/algorithm/ppo/ppo.py")
    imp = _load("xt.algorithm.impala.impala_opt", "xt/algorithm/impala/impala_opt.py")
    imp0 = _load("xt.algorithm.impala.impala", "xt/algorithm/impala/impala.py")
    return ppo.PPO, imp.IMPALAOpt, imp0.IMPALA


def ppo_inputs(seed=0):
    rng = np.random.default_rng(seed)
    trajs = []
    for t in (5, 7, 4):                      # ragged: variable-length episodes (CartPole-style)
        trajs.append({"cur_state": rng.integers(0, 256, (t, 6, 6, 2)).astype(np.uint8),
                      "action": rng.integers(0, 4, t).astype(np.int32),
                      "logp": (-np.abs(rng.standard_normal((t, 1)))).astype(np.float32),
                      "adv": rng.standard_normal((t, 1)),                          # float64 as the agent produces it
                      "old_value": rng.standard_normal((t, 1)).astype(np.float32),
                      "target_value": rng.standard_normal((t, 1))})
    return trajs


def impala_inputs(seed=1):
    rng = np.random.default_rng(seed)
    msgs = []
    for _ in range(2):
        n = 10
        msgs.append({"cur_state": rng.integers(0, 256, (n, 6, 6, 2)).astype(np.uint8),
                     "logit": rng.standard_normal((n, 3)).astype(np.float32),
                     "action": rng.integers(0, 3, n).astype(np.int32),
                     "done": [bool(x) for x in (rng.random(n) < 0.2)],             # python lists, as the agent ships
                     "reward": [float(x) for x in rng.choice([-1.0, 0.0, 1.0], n)]})
    return msgs


def impala_plain_inputs(seed=2, episode_len=6, a_dim=3):
    """Two fragments of episode_len transitions (episode_len + 1 states each) as the IMPALA agent ships them:
    one-hot ``real_action``, behaviour probabilities in ``action``."""
    rng = np.random.default_rng(seed)
    msgs = []
    for _ in range(2):
        t = episode_len
        beh = rng.random((t, a_dim)) + 0.1
        msgs.append({"cur_state": rng.integers(0, 256, (t + 1, 6, 6, 2)).astype(np.uint8),
                     "real_action": np.eye(a_dim, dtype=np.float32)[rng.integers(0, a_dim, t)],
                     "reward": [float(x) for x in rng.choice([-1.0, 0.0, 1.0], t)],
                     "done": [bool(x) for x in (rng.random(t) < 0.25)],
                     "action": (beh / beh.sum(-1, keepdims=True)).astype(np.float32)})
    return msgs


IMPALA_PLAIN_CFG = ({"actor": {"model_name": "RecordingModel", "state_dim": [6, 6, 2], "action_dim": 3}},
                    {"instance_num": 2, "agent_num": 1, "prepare_times_per_train": 2, "BATCH_SIZE": 5,
                     "episode_len": 6})
PPO_CFG = ({"actor": {"model_name": "RecordingModel", "state_dim": [6, 6, 2], "action_dim": 4}},
           {"instance_num": 3, "agent_num": 1})
IMPALA_CFG = ({"actor": {"model_name": "RecordingModel", "state_dim": [6, 6, 2], "action_dim": 3}},
              {"instance_num": 2, "agent_num": 1, "prepare_times_per_train": 2, "BATCH_SIZE": 8})


def pack_calls(calls, prefix, out):
    out[prefix + "_ncalls"] = np.int64(len(calls))
    for i, (state, label) in enumerate(calls):
        st = state if isinstance(state, list) else [state]
        out["%s_%d_nstate" % (prefix, i)] = np.int64(len(st))
        out["%s_%d_state_is_list" % (prefix, i)] = np.bool_(isinstance(state, list))
        for j, a in enumerate(st):
            out["%s_%d_state_%d" % (prefix, i, j)] = a
        out["%s_%d_nlabel" % (prefix, i)] = np.int64(len(label))
        for j, a in enumerate(label):
            out["%s_%d_label_%d" % (prefix, i, j)] = a


def main():
    PPO, IMPALAOpt, IMPALA = load_reference_algorithms()
    os.makedirs(OUT, exist_ok=True)
    out = {}
    alg = PPO(*PPO_CFG)
    assert alg.async_flag is False and alg.prepare_data_times == 3
    for tr in ppo_inputs():
        alg.prepare_data(tr)
    out["loss"] = np.float64(alg.train())
    pack_calls(alg.actor.calls, "train", out)
    out["lists_cleared"] = np.bool_(alg.obs == [] and alg.adv == [])
    np.savez_compressed(os.path.join(OUT, "alg_ppo.npz"), **out)
    out = {}
    alg = IMPALAOpt(*IMPALA_CFG)
    assert alg.async_flag is False and alg.prepare_data_times == 2
    for m in impala_inputs():
        alg.prepare_data(m)
    out["loss"] = np.float64(alg.train())
    pack_calls(alg.actor.calls, "train", out)
    out["lists_cleared"] = np.bool_(alg.states == [] and alg.rewards == [])
    np.savez_compressed(os.path.join(OUT, "alg_impala_opt.npz"), **out)
    print("wrote alg_ppo.npz (%d train call) and alg_impala_opt.npz (%d train calls)"
          % (1, int(out["train_ncalls"])))
    out = {}
    alg = IMPALA(*IMPALA_PLAIN_CFG)
    assert alg.async_flag is False and alg.episode_len == 6
    for m in impala_plain_inputs():
        alg.prepare_data(m)
    out["loss"] = np.float64(alg.train())
    pack_calls(alg.actor.calls, "train", out)
    out["pred_p"], out["pred_v"], out["pred_state"] = alg.actor.pred[0], alg.actor.pred[1], alg.actor.pred_state
    out["lists_cleared"] = np.bool_(alg.state == [] and alg.rewards == [])
    single = alg.predict(np.zeros((6, 6, 2), np.uint8))
    out["single_pred_p"], out["single_pred_v"] = single[0], single[1]
    np.savez_compressed(os.path.join(OUT, "alg_impala.npz"), **out)
    print("wrote alg_impala.npz (%d train calls)" % int(out["train_ncalls"]))


if __name__ == "__main__":
    main()
