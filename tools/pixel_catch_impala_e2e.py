#!/usr/bin/env python
"""Closed-loop IMPALA (v-trace on the GPU) on PIXELS through ``IMPALAOpt`` + ``ImpalaCnnOpt`` with
examples/breakout_impala.yaml's sections (84x84x4 uint8, state_std 255, A = 4, LR 5e-4, grad_norm_clip 40,
sample_batch_step 128, BATCH_SIZE 512, prepare_times_per_train 1, train_per_checkpoint 1, env_num 32) and the synthetic
Atari-shaped catch game of tools/pixel_catch_e2e.py in place of the emulator.

    explorers   32 games in lock step, played by a SECOND ImpalaCnnOpt that receives the learner's weights by name; one
                message per game and round in the format of ``AtariImpalaOpt.get_trajectory``
                (xt/agent/impala/atari_impala_opt.py:96-110): 128 uint8 stacks, the behaviour LOGITS, actions, done, reward
    learner     one ``prepare_data`` + ``train`` per message (prepare_times_per_train 1): a 128-frame SGD step with the
                v-trace targets computed on the GPU from the behaviour logits shipped with the frames.  The 32 messages of
                a round were all played with the weights of the round's start, so the k-th of them is k trains stale --
                the off-policy lag of 32 asynchronous actors, which is what v-trace's truncated importance weights are for

Prints the mean episode return per round (random play: about -0.7; perfect play: +1).  Usage (GPU box):
    python tools/pixel_catch_impala_e2e.py [rounds]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from pixel_catch_e2e import SCREEN, CatchVec  # noqa: E402

MODEL_CONFIG = dict(LR=0.0005, grad_norm_clip=40.0, sample_batch_step=128)              # breakout_impala.yaml:29-35
ALG_CONFIG = dict(BATCH_SIZE=512, prepare_times_per_train=1, train_per_checkpoint=1)    # breakout_impala.yaml:3-7
ENV_NUM, T = 32, 128


def run(rounds=40, seed=0, verbose=True):
    from xingtian_amd.algorithm import alg_builder
    from xingtian_amd.model import model_builder
    info = {"model_name": "ImpalaCnnOpt", "state_dim": [84, 84, 4], "action_dim": 4, "input_dtype": "uint8",
            "state_mean": 0.0, "state_std": 255.0, "type": "learner"}
    learner = alg_builder("IMPALAOpt", {"actor": dict(info, model_config=dict(MODEL_CONFIG, SEED=seed))},
                          dict(ALG_CONFIG, instance_num=ENV_NUM, agent_num=1))
    actor = model_builder(dict(info, max_batch=512, model_config=dict(MODEL_CONFIG, SEED=seed + 1)))
    actor.set_weights(learner.get_weights())
    env = CatchVec(ENV_NUM, seed)
    running = np.zeros(ENV_NUM)
    st = np.empty((ENV_NUM, T, SCREEN, SCREEN, 4), np.uint8)
    logit = np.empty((ENV_NUM, T, 4), np.float32)
    act = np.empty((ENV_NUM, T), np.int32)
    rew, don = np.empty((ENV_NUM, T)), np.empty((ENV_NUM, T), bool)
    curve = []
    t_env = t_learn = 0.0
    for rnd in range(rounds):
        t0 = time.time()
        finished, losses = [], []
        for t in range(T):
            st[:, t] = env.obs
            logit[:, t], _, act[:, t] = actor.predict(env.obs)
            rew[:, t], don[:, t] = env.step(act[:, t])
            running += rew[:, t]
            for i in np.nonzero(don[:, t])[0]:
                finished.append(running[i])
                running[i] = 0.0
        t1 = time.time()
        for i in range(ENV_NUM):
            learner.prepare_data({"cur_state": st[i], "logit": logit[i], "action": act[i], "done": list(don[i]),
                                  "reward": list(rew[i])})
            losses.append(learner.train(episode_num=rnd * ENV_NUM + i))
        actor.set_weights(learner.get_weights())
        t2 = time.time()
        t_env, t_learn = t_env + (t1 - t0), t_learn + (t2 - t1)
        mean_ret = float(np.mean(finished))
        curve.append(mean_ret)
        if verbose:
            print("round %3d  env-frames %7d  trains %4d  episodes %3d  mean return %6.3f  loss %9.4f  (rollout %.2f s, learner %.3f s)"
                  % (rnd, (rnd + 1) * ENV_NUM * T, (rnd + 1) * ENV_NUM, len(finished), mean_ret, float(np.mean(losses)),
                     t1 - t0, t2 - t1), flush=True)
    if verbose:
        print("rollouts %.1f s, learner (prepare_data + train per message, weights per round) %.2f s for %d env-frames"
              % (t_env, t_learn, rounds * ENV_NUM * T))
    return curve


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
