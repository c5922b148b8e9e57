#!/usr/bin/env python
"""Closed-loop PPO on CartPole through the plugin classes (BASELINE configs[0]: examples/cartpole_ppo.yaml).

The whole XingTian data path of this repo in one process, at toy scale:

    explorers   ``model_builder(PpoMlp)`` built WITHOUT a GPU role -> the inference-only numpy replica
                (xingtian_amd/model/cpu_net.py): ``predict(state)`` -> action, log-prob, value
    environment a numpy CartPole (the classic cart-pole equations: gravity 9.8, cart 1.0 kg, pole 0.1 kg / 0.5 m
                half-length, 10 N pushes, 20 ms Euler steps, failure at |x| > 2.4 or |theta| > 12 degrees, 200-step cap)
    learner     ``alg_builder("PPO")`` with ``type: learner`` -> HIP kernels: trajectories arrive with raw
                value / reward / done, GAE runs on the GPU (xt_gae_f64), ``train()`` = NUM_SGD_ITER x minibatches in one
                C call, ``get_weights()`` -> name-keyed dict -> ``set_weights`` of the explorers' replica

Prints the mean episode return per update (the "reward curve"); ``run()`` returns it.  Usage (GPU box):
    python tools/cartpole_e2e.py [updates]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODEL_CONFIG = dict(BATCH_SIZE=200, CRITIC_LOSS_COEF=1.0, ENTROPY_LOSS=0.01, LR=0.0003, LOSS_CLIPPING=0.2,
                    MAX_GRAD_NORM=5.0, NUM_SGD_ITER=8, SUMMARY=False, VF_SHARE_LAYERS=False, activation="tanh",
                    hidden_sizes=[64, 64], action_type="Categorical")       # examples/cartpole_ppo.yaml:26-38
ENV_NUM, MAX_STEPS = 10, 200                                                 # env_num: 10, agent_config.max_steps: 200


class CartPole(object):
    GRAVITY, M_CART, M_POLE, HALF_LEN, FORCE, TAU = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    X_LIMIT, THETA_LIMIT, STEP_CAP = 2.4, 12.0 * 2.0 * np.pi / 360.0, 200

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.reset()

    def reset(self):
        self.state = self.rng.uniform(-0.05, 0.05, 4)
        self.steps = 0
        return self.state.astype(np.float32)

    def step(self, action):
        x, x_dot, th, th_dot = self.state
        force = self.FORCE if action == 1 else -self.FORCE
        total = self.M_CART + self.M_POLE
        pml = self.M_POLE * self.HALF_LEN
        temp = (force + pml * th_dot * th_dot * np.sin(th)) / total
        th_acc = (self.GRAVITY * np.sin(th) - np.cos(th) * temp) / (
            self.HALF_LEN * (4.0 / 3.0 - self.M_POLE * np.cos(th) ** 2 / total))
        x_acc = temp - pml * th_acc * np.cos(th) / total
        self.state = np.array([x + self.TAU * x_dot, x_dot + self.TAU * x_acc, th + self.TAU * th_dot,
                               th_dot + self.TAU * th_acc])
        self.steps += 1
        failed = abs(self.state[0]) > self.X_LIMIT or abs(self.state[2]) > self.THETA_LIMIT
        done = bool(failed or self.steps >= self.STEP_CAP)
        return self.state.astype(np.float32), 1.0, done


def run(updates=30, seed=0, verbose=True):
    from xingtian_amd.algorithm import alg_builder
    from xingtian_amd.model import model_builder
    info = {"model_name": "PpoMlp", "state_dim": [4], "action_dim": 2, "input_dtype": "float32"}
    learner = alg_builder("PPO", {"actor": dict(info, type="learner", model_config=dict(MODEL_CONFIG, SEED=seed))},
                          {"instance_num": ENV_NUM, "agent_num": 1})
    actor = model_builder(dict(info, model_config=dict(MODEL_CONFIG, SEED=seed + 1, DEVICE="cpu")))   # explorer side
    assert actor.net.inference_only and not learner.actor.net.inference_only
    actor.set_weights(learner.get_weights())
    envs = [CartPole(seed * 1000 + i) for i in range(ENV_NUM)]
    states = [e.reset() for e in envs]
    running = [0.0] * ENV_NUM
    curve = []
    for upd in range(updates):
        finished = []
        for i, env in enumerate(envs):
            tr = {"cur_state": [], "action": [], "logp": [], "value": [], "reward": [], "done": []}
            s = states[i]
            for _ in range(MAX_STEPS):
                action, logp, value = actor.predict(s.reshape(1, 4))
                s2, r, done = env.step(int(action[0]))
                tr["cur_state"].append(s); tr["action"].append(action[0]); tr["logp"].append(logp[0])
                tr["value"].append(value[0]); tr["reward"].append(r); tr["done"].append(done)
                running[i] += r
                if done:
                    finished.append(running[i])
                    running[i] = 0.0
                    s2 = env.reset()
                s = s2
            states[i] = s
            _, _, last_v = actor.predict(s.reshape(1, 4))
            tr["value"].append(last_v[0])
            learner.prepare_data({"cur_state": np.asarray(tr["cur_state"], np.float32),
                                  "action": np.asarray(tr["action"], np.int32),
                                  "logp": np.asarray(tr["logp"], np.float32).reshape(-1, 1),
                                  "value": np.asarray(tr["value"], np.float32).reshape(-1, 1),
                                  "reward": np.asarray(tr["reward"], np.float64), "done": np.asarray(tr["done"], bool)})
        loss = learner.train(episode_num=upd)
        actor.set_weights(learner.get_weights())            # the weight publish of the learner loop
        mean_ret = float(np.mean(finished)) if finished else float("nan")
        curve.append(mean_ret)
        if verbose:
            print("update %3d  env-steps %6d  episodes %3d  mean return %6.1f  loss %8.4f"
                  % (upd, (upd + 1) * ENV_NUM * MAX_STEPS, len(finished), mean_ret, loss), flush=True)
    return curve


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
