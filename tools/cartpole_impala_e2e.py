#!/usr/bin/env python
"""Closed-loop IMPALA on CartPole through the plugin classes (examples/cartpole_impala.yaml: ``IMPALA`` + ``ImpalaMlp``,
BATCH_SIZE 800, episode_len 200, prepare_times_per_train 2, train_per_checkpoint 2, env_num 10).

The second algorithm family of the hot path end to end, at toy scale and in one process:

    explorers   ``model_builder(ImpalaMlp)`` built WITHOUT a GPU role (numpy replica): ``predict`` -> action
                probabilities + value; the action is drawn from them (xt/agent/ppo/cartpole_ppo.py:58)
    fragments   what ``CartpoleImpala.data_proc`` ships (xt/agent/impala/cartpole_impala.py:46-58): episode_len + 1
                states, one-hot ``real_action``, the behaviour PROBABILITIES in ``action``, reward, done -- the
                environment is reset inside a fragment when an episode ends (:89-91)
    learner     ``alg_builder("IMPALA")`` with ``type: learner``: forward over all stored states on the GPU, v-trace from
                probabilities on the host (the reference's own index convention, algorithm/impala/impala.py), one
                Keras-form ``fit`` epoch (minibatches of 128, impala_loss + 0.5 mse, tf.keras Adam) on HIP kernels
    publish     every ``train_per_checkpoint`` trains the explorers take ``get_weights()``; fragments collected before
                that were produced by an older policy -- the lag v-trace corrects for

Prints the mean episode return per round of env_num fragments; ``run()`` returns the curve.  Usage (GPU box):
    python tools/cartpole_impala_e2e.py [rounds]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from cartpole_e2e import CartPole  # noqa: E402  (the numpy cart-pole of the PPO closed loop)

ALG_CONFIG = dict(train_per_checkpoint=2, prepare_times_per_train=2, BATCH_SIZE=800, episode_len=200)   # yaml:3-8
ENV_NUM, MAX_STEPS, ACTION_DIM = 10, 200, 2                                                             # yaml:18-21,31


def collect_fragment(env, state, actor, rng, running, finished):
    """One ``run_one_episode`` of the reference agent: MAX_STEPS transitions, resets inside the fragment."""
    states, onehot, probs, rewards, dones = [], [], [], [], []
    eye = np.eye(ACTION_DIM)
    for _ in range(MAX_STEPS):
        prob = actor.predict([state.reshape(1, 4), np.zeros((1, 1))])[0][0]
        a = int(rng.choice(ACTION_DIM, p=np.nan_to_num(prob)))
        nxt, r, done = env.step(a)
        states.append(state); onehot.append(eye[a]); probs.append(prob); rewards.append(r); dones.append(done)
        running[0] += r
        if done:
            finished.append(running[0])
            running[0] = 0.0
            nxt = env.reset()
        state = nxt
    states.append(state)                                  # "last_state": episode_len + 1 states per fragment
    return state, {"cur_state": np.asarray(states, np.float32), "real_action": np.asarray(onehot),
                   "action": np.asarray(probs, np.float32), "reward": np.asarray(rewards, np.float64),
                   "done": np.asarray(dones, bool)}


def run(rounds=60, seed=0, verbose=True):
    from xingtian_amd.algorithm import alg_builder
    from xingtian_amd.model import model_builder
    info = {"model_name": "ImpalaMlp", "state_dim": [4], "action_dim": ACTION_DIM, "input_dtype": "float32"}
    np.random.seed(seed)                                  # model.fit(shuffle=True) draws from numpy's global generator
    learner = alg_builder("IMPALA", {"actor": dict(info, type="learner", model_config={"SEED": seed})},
                          dict(ALG_CONFIG, instance_num=ENV_NUM, agent_num=1))
    actor = model_builder(dict(info, model_config={"SEED": seed + 1, "DEVICE": "cpu"}))
    assert actor.net.inference_only and not learner.actor.net.inference_only
    actor.set_weights(learner.get_weights())
    rng = np.random.default_rng(seed)
    envs = [CartPole(seed * 1000 + i) for i in range(ENV_NUM)]
    states = [e.reset() for e in envs]
    running = [[0.0] for _ in range(ENV_NUM)]
    curve, trains = [], 0
    for rnd in range(rounds):
        finished, losses = [], []
        for i, env in enumerate(envs):
            states[i], frag = collect_fragment(env, states[i], actor, rng, running[i], finished)
            learner.prepare_data(frag)
            if (i + 1) % ALG_CONFIG["prepare_times_per_train"] == 0:
                losses.append(learner.train(episode_num=trains))
                trains += 1
                if trains % ALG_CONFIG["train_per_checkpoint"] == 0:
                    actor.set_weights(learner.get_weights())
        mean_ret = float(np.mean(finished)) if finished else float("nan")
        curve.append(mean_ret)
        if verbose:
            print("round %3d  env-steps %6d  trains %4d  episodes %3d  mean return %6.1f  loss %9.4f"
                  % (rnd, (rnd + 1) * ENV_NUM * MAX_STEPS, trains, len(finished), mean_ret, float(np.mean(losses))),
                  flush=True)
    return curve


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60)
