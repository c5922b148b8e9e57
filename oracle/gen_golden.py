"""Oracle (test infrastructure): generate golden GAE vectors by EXECUTING the reference.

Runs only in the authoring container (needs /root/reference, read-only).  The
reference file ``xt/agent/ppo/ppo.py`` is loaded with importlib after stubbing
the three modules it imports (``xt.agent.Agent``, ``xt.agent.ppo.default_config``
and ``zeus.common.util.register.Registers``) so that ``PPO.data_proc`` (:77-106)
runs unmodified on seeded synthetic trajectories.  The outputs are committed as
``tests/golden/gae_*.npz``; nothing at test/bench time reads /root/reference.

Usage:  python oracle/gen_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_reference_ppo_agent():
    class _Agent(object):
        def __init__(self, *a, **k):
            pass

    class _Stub(object):
        def __call__(self, cls):
            return cls

    class _Registers(object):
        agent = _Stub()

    mods = {
        "xt": types.ModuleType("xt"),
        "xt.agent": types.ModuleType("xt.agent"),
        "xt.agent.ppo": types.ModuleType("xt.agent.ppo"),
        "xt.agent.ppo.default_config": types.ModuleType("xt.agent.ppo.default_config"),
        "zeus": types.ModuleType("zeus"),
        "zeus.common": types.ModuleType("zeus.common"),
        "zeus.common.util": types.ModuleType("zeus.common.util"),
        "zeus.common.util.register": types.ModuleType("zeus.common.util.register"),
    }
    mods["xt.agent"].Agent = _Agent
    # the reference's own constants, read from its file (GAMMA/LAM)
    cfg = {}
    with open(os.path.join(REF, "xt/agent/ppo/default_config.py")) as f:
        exec(f.read(), cfg)
    mods["xt.agent.ppo.default_config"].GAMMA = cfg["GAMMA"]
    mods["xt.agent.ppo.default_config"].LAM = cfg["LAM"]
    mods["zeus.common.util.register"].Registers = _Registers
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        spec = importlib.util.spec_from_file_location("_ref_ppo_agent", os.path.join(REF, "xt/agent/ppo/ppo.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod.PPO, cfg["GAMMA"], cfg["LAM"]


def make_traj(rng, t_len, done_mode):
    value = rng.standard_normal((t_len + 1, 1)).astype(np.float32)
    reward = rng.choice([-1.0, 0.0, 1.0], size=t_len, p=[0.05, 0.9, 0.05])
    if done_mode == "none":
        done = np.zeros(t_len, bool)
    elif done_mode == "all":
        done = np.ones(t_len, bool)
    elif done_mode == "first":
        done = np.zeros(t_len, bool); done[0] = True
    elif done_mode == "last":
        done = np.zeros(t_len, bool); done[-1] = True
    elif done_mode == "dense":
        done = rng.random(t_len) < 0.3
    else:
        done = rng.random(t_len) < 0.01
    return value, reward, done


def run_reference(ppo_cls, value, reward, done):
    agent = ppo_cls.__new__(ppo_cls)
    t_len = len(reward)
    agent.trajectory = {
        "cur_state": [np.zeros((1,), np.uint8) for _ in range(t_len)],
        "action": [np.int32(0) for _ in range(t_len)],
        "logp": [np.zeros((1,), np.float32) for _ in range(t_len)],
        "value": [value[i] for i in range(t_len + 1)],          # [1] float32 each (predict_val[2][0])
        "reward": [float(r) for r in reward],                    # python floats
        "done": [bool(d) for d in done],
    }
    agent.data_proc()
    tr = agent.trajectory
    return tr["adv"], tr["old_value"], tr["target_value"]


def main():
    ppo_cls, gamma, lam = load_reference_ppo_agent()
    os.makedirs(OUT, exist_ok=True)
    cases = [("bernoulli", 128, 0), ("bernoulli", 128, 1), ("bernoulli", 128, 2), ("none", 128, 3),
             ("all", 128, 4), ("first", 128, 5), ("last", 128, 6), ("dense", 200, 7), ("dense", 1, 8),
             ("bernoulli", 2, 9), ("dense", 37, 10)]
    for mode, t_len, seed in cases:
        rng = np.random.default_rng(seed)
        value, reward, done = make_traj(rng, t_len, mode)
        adv, old_v, tgt = run_reference(ppo_cls, value.copy(), reward.copy(), done.copy())
        assert adv.dtype == np.float64 and tgt.dtype == np.float64 and old_v.dtype == np.float32
        name = "gae_{}_T{}_s{}.npz".format(mode, t_len, seed)
        np.savez(os.path.join(OUT, name), value=value, reward=reward, done=done, adv=adv,
                 old_value=old_v, target_value=tgt, gamma=gamma, lam=lam)
        print("wrote", name, adv.shape, adv.dtype)


if __name__ == "__main__":
    main()
