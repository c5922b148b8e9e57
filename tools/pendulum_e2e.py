#!/usr/bin/env python
"""Closed-loop continuous-action PPO on a numpy Pendulum through the plugin classes (examples/pendulum_ppo.yaml:
``PPO`` + ``PpoMlp``, state 3, action 1, DiagGaussian policy with the free ``pi_logstd`` variable).

Same data path as tools/cartpole_e2e.py (CPU-replica explorers -> raw trajectories -> GAE on the learner GPU ->
``train()`` -> weights by name), with the Gaussian head: the explorers draw ``mean + exp(pi_logstd) * N(0, 1)`` and ship
the float action and its log-probability (xt/model/tf_dist.py:66-87).

Environment: the classic torque-limited pendulum swing-up (g 10, m 1, l 1, 50 ms steps, torque clipped to [-2, 2],
speed to [-8, 8], reward -(angle^2 + 0.1 speed^2 + 0.001 torque^2), 200-step episodes).  Prints the mean episode
return per update; ``run()`` returns the curve.  Usage (GPU box):  python tools/pendulum_e2e.py [updates]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODEL_CONFIG = dict(BATCH_SIZE=200, CRITIC_LOSS_COEF=1.0, ENTROPY_LOSS=0.01, LR=0.0003, LOSS_CLIPPING=0.2,
                    MAX_GRAD_NORM=5.0, NUM_SGD_ITER=8, SUMMARY=False, VF_SHARE_LAYERS=False, activation="tanh",
                    hidden_sizes=[64, 64], action_type="DiagGaussian")      # examples/pendulum_ppo.yaml:26-37
ENV_NUM, MAX_STEPS = 10, 200                                                 # env_num: 10, agent_config.max_steps: 200


class Pendulum(object):
    MAX_SPEED, MAX_TORQUE, DT, G, M, L, STEP_CAP = 8.0, 2.0, 0.05, 10.0, 1.0, 1.0, 200

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.reset()

    def _obs(self):
        return np.array([np.cos(self.th), np.sin(self.th), self.thdot], np.float32)

    def reset(self):
        self.th, self.thdot = self.rng.uniform(-np.pi, np.pi), self.rng.uniform(-1.0, 1.0)
        self.steps = 0
        return self._obs()

    def step(self, action):
        u = float(np.clip(action, -self.MAX_TORQUE, self.MAX_TORQUE))
        ang = ((self.th + np.pi) % (2.0 * np.pi)) - np.pi
        cost = ang * ang + 0.1 * self.thdot * self.thdot + 0.001 * u * u
        self.thdot = self.thdot + (-3.0 * self.G / (2.0 * self.L) * np.sin(self.th + np.pi)
                                   + 3.0 / (self.M * self.L * self.L) * u) * self.DT
        self.th = self.th + self.thdot * self.DT
        self.thdot = float(np.clip(self.thdot, -self.MAX_SPEED, self.MAX_SPEED))
        self.steps += 1
        return self._obs(), -cost, self.steps >= self.STEP_CAP


def run(updates=300, seed=0, verbose=True, model_config=None):
    from xingtian_amd.algorithm import alg_builder
    from xingtian_amd.model import model_builder
    mc = dict(MODEL_CONFIG, **(model_config or {}))
    info = {"model_name": "PpoMlp", "state_dim": [3], "action_dim": 1, "input_dtype": "float32"}
    learner = alg_builder("PPO", {"actor": dict(info, type="learner", model_config=dict(mc, SEED=seed))},
                          {"instance_num": ENV_NUM, "agent_num": 1})
    actor = model_builder(dict(info, model_config=dict(mc, SEED=seed + 1, DEVICE="cpu")))
    assert actor.net.inference_only and not learner.actor.net.inference_only
    actor.set_weights(learner.get_weights())
    envs = [Pendulum(seed * 1000 + i) for i in range(ENV_NUM)]
    states = [e.reset() for e in envs]
    running = [0.0] * ENV_NUM
    curve = []
    for upd in range(updates):
        finished = []
        for i, env in enumerate(envs):
            tr = {"cur_state": [], "action": [], "logp": [], "value": [], "reward": [], "done": []}
            s = states[i]
            for _ in range(MAX_STEPS):
                action, logp, value = actor.predict(s.reshape(1, 3))
                s2, r, done = env.step(action[0, 0])
                tr["cur_state"].append(s); tr["action"].append(action[0]); tr["logp"].append(logp[0])
                tr["value"].append(value[0]); tr["reward"].append(r); tr["done"].append(done)
                running[i] += r
                if done:
                    finished.append(running[i])
                    running[i] = 0.0
                    s2 = env.reset()
                s = s2
            states[i] = s
            _, _, last_v = actor.predict(s.reshape(1, 3))
            tr["value"].append(last_v[0])
            learner.prepare_data({"cur_state": np.asarray(tr["cur_state"], np.float32),
                                  "action": np.asarray(tr["action"], np.float32).reshape(-1, 1),
                                  "logp": np.asarray(tr["logp"], np.float32).reshape(-1, 1),
                                  "value": np.asarray(tr["value"], np.float32).reshape(-1, 1),
                                  "reward": np.asarray(tr["reward"], np.float64), "done": np.asarray(tr["done"], bool)})
        loss = learner.train(episode_num=upd)
        actor.set_weights(learner.get_weights())
        mean_ret = float(np.mean(finished)) if finished else float("nan")
        curve.append(mean_ret)
        if verbose:
            print("update %3d  env-steps %7d  episodes %3d  mean return %8.1f  loss %10.4f  logstd %6.3f"
                  % (upd, (upd + 1) * ENV_NUM * MAX_STEPS, len(finished), mean_ret, loss,
                     float(learner.get_weights()["pi_logstd"].reshape(-1)[0])), flush=True)
    return curve


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 300)
