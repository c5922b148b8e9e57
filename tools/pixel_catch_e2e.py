#!/usr/bin/env python
"""Closed-loop PPO on PIXELS through the headline network (examples/breakout_ppo.yaml's model section: ``PpoCnn``,
84x84x4 uint8 frame stacks, A = 4, BATCH_SIZE 320, NUM_SGD_ITER 4, LR 2.5e-4, clip 0.1, entropy 0.003, hidden [256];
max_steps 128), with a synthetic Atari-shaped game in place of the emulator (no ROMs in this image).

    game        "catch": a 4x4 ball falls 4 px per step from a random column of an 84x84 screen, a 12x4 paddle on the
                bottom rows moves 4 px left / right (Breakout's action set: 0 noop, 1 fire = noop, 2 right, 3 left);
                +1 when the ball lands on the paddle, -1 otherwise, episode over (19 steps); observations are stacks of
                the last four frames, uint8 -- the byte format, shapes and reward scale of the Atari agents
                (xt/agent/ppo/atari_ppo.py)
    explorers   ENV_NUM games stepped in lock step; their policy is a SECOND PpoCnn instance that receives the learner's
                weights by name after every update and predicts the whole batch of stacks in one HIP forward (the
                reference runs one TF session per explorer process; the numpy replica works too but needs ~5 ms per frame)
    learner     ``alg_builder("PPO")``, ``type: learner``: ENV_NUM raw trajectories of 128 uint8 stacks per update
                through ``prepare_data`` (pinned staging, H2D), GAE on the GPU, 4 epochs x 320-row minibatches in one
                replayed hipGraph -- the path ``bench.py`` times, with real data dependence between updates

Prints the mean episode return per update (random play: about -0.6; perfect play: +1).  Usage (GPU box):
    python tools/pixel_catch_e2e.py [updates] [env_num]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODEL_CONFIG = dict(BATCH_SIZE=320, CRITIC_LOSS_COEF=1.0, ENTROPY_LOSS=0.003, LOSS_CLIPPING=0.1, LR=0.00025,
                    MAX_GRAD_NORM=5.0, NUM_SGD_ITER=4, SUMMARY=False, VF_SHARE_LAYERS=True, activation="relu",
                    hidden_sizes=[256], action_type="Categorical")            # examples/breakout_ppo.yaml:19-31
MAX_STEPS, SCREEN, BALL, PADDLE_W, PADDLE_ROW, STEP_PX = 128, 84, 4, 12, 78, 4


class CatchVec(object):
    """n independent games; ``obs`` is the [n, 84, 84, 4] uint8 stack of the last four frames (newest last)."""

    def __init__(self, n, seed):
        self.n, self.rng = n, np.random.default_rng(seed)
        self.bx, self.by, self.px = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int64)
        self.obs = np.zeros((n, SCREEN, SCREEN, 4), np.uint8)
        for i in range(n):
            self._reset(i)

    def _frame(self, i):
        f = np.zeros((SCREEN, SCREEN), np.uint8)
        f[self.by[i]:self.by[i] + BALL, self.bx[i]:self.bx[i] + BALL] = 255
        f[PADDLE_ROW:PADDLE_ROW + 4, self.px[i]:self.px[i] + PADDLE_W] = 200
        return f

    def _reset(self, i):
        self.bx[i] = STEP_PX * self.rng.integers(0, (SCREEN - BALL) // STEP_PX + 1)
        self.by[i] = 0
        self.px[i] = STEP_PX * self.rng.integers(0, (SCREEN - PADDLE_W) // STEP_PX + 1)
        self.obs[i] = self._frame(i)[:, :, None]                 # a fresh episode starts with four copies of its frame

    def step(self, actions):
        """-> (reward [n] float64, done [n] bool); ``obs`` then holds the next state of every game."""
        reward, done = np.zeros(self.n), np.zeros(self.n, bool)
        move = np.where(actions == 2, STEP_PX, np.where(actions == 3, -STEP_PX, 0))
        self.px = np.clip(self.px + move, 0, SCREEN - PADDLE_W)
        self.by = self.by + STEP_PX
        for i in range(self.n):
            if self.by[i] + BALL > PADDLE_ROW:                   # the ball reached the paddle rows
                hit = self.px[i] - BALL < self.bx[i] < self.px[i] + PADDLE_W
                reward[i], done[i] = (1.0 if hit else -1.0), True
                self._reset(i)
            else:
                self.obs[i, :, :, :3] = self.obs[i, :, :, 1:]
                self.obs[i, :, :, 3] = self._frame(i)
        return reward, done


def run(updates=60, env_num=32, seed=0, verbose=True):
    from xingtian_amd.algorithm import alg_builder
    from xingtian_amd.model import model_builder
    info = {"model_name": "PpoCnn", "state_dim": [84, 84, 4], "action_dim": 4, "input_dtype": "uint8", "type": "learner"}
    learner = alg_builder("PPO", {"actor": dict(info, model_config=dict(MODEL_CONFIG, SEED=seed))},
                          {"instance_num": env_num, "agent_num": 1})
    actor = model_builder(dict(info, model_config=dict(MODEL_CONFIG, SEED=seed + 1)))
    actor.set_weights(learner.get_weights())
    env = CatchVec(env_num, seed)
    running = np.zeros(env_num)
    curve = []
    t_env = t_learn = 0.0
    # per-explorer buffers, time-major inside an explorer: message i is the contiguous block [i]
    st = np.empty((env_num, MAX_STEPS, SCREEN, SCREEN, 4), np.uint8)
    act = np.empty((env_num, MAX_STEPS), np.int32)
    logp = np.empty((env_num, MAX_STEPS, 1), np.float32)
    val = np.empty((env_num, MAX_STEPS + 1, 1), np.float32)
    rew, don = np.empty((env_num, MAX_STEPS)), np.empty((env_num, MAX_STEPS), bool)
    for upd in range(updates):
        t0 = time.time()
        finished = []
        for t in range(MAX_STEPS):
            st[:, t] = env.obs
            act[:, t], logp[:, t], val[:, t] = actor.predict(env.obs)
            rew[:, t], don[:, t] = env.step(act[:, t])
            running += rew[:, t]
            for i in np.nonzero(don[:, t])[0]:
                finished.append(running[i])
                running[i] = 0.0
        val[:, MAX_STEPS] = actor.predict(env.obs)[2]
        t1 = time.time()
        for i in range(env_num):                               # one message per explorer, as the broker delivers them
            learner.prepare_data({"cur_state": st[i], "action": act[i], "logp": logp[i], "value": val[i],
                                  "reward": rew[i], "done": don[i]})
        loss = learner.train(episode_num=upd)
        actor.set_weights(learner.get_weights())
        t2 = time.time()
        t_env, t_learn = t_env + (t1 - t0), t_learn + (t2 - t1)
        mean_ret = float(np.mean(finished))
        curve.append(mean_ret)
        if verbose:
            print("update %3d  env-frames %7d  episodes %3d  mean return %6.3f  loss %8.4f  (rollout %.2f s, learner %.3f s)"
                  % (upd, (upd + 1) * env_num * MAX_STEPS, len(finished), mean_ret, loss, t1 - t0, t2 - t1), flush=True)
    if verbose:
        print("rollouts %.1f s, learner (prepare_data + train + weights) %.2f s for %d env-frames"
              % (t_env, t_learn, updates * env_num * MAX_STEPS))
    return curve


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 32)
