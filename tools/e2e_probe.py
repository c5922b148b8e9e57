#!/usr/bin/env python
"""Host-side probes of the plugin path on the GPU box: (1) H2D rate of pinned memory vs copy size, (2) staging + H2D of
a 32-trajectory rollout vs xt_stage_rows' ship size and thread count, (3) cProfile of the PPO / IMPALA plugin loops.
usage: python tools/e2e_probe.py [h2d] [stage] [ppo] [impala]"""
import cProfile
import ctypes
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xingtian_amd import ingest, lib as L  # noqa: E402

what = sys.argv[1:] or ["h2d", "stage", "ppo", "impala"]
dev = torch.device("cuda:0")
h = L.load()

if "h2d" in what:
    total = 115 << 20
    pin = torch.empty((total,), dtype=torch.uint8, pin_memory=True)
    dst = torch.empty((total,), dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream()
    for piece in (256 << 10, 1 << 20, 2 << 20, 3612672, 8 << 20, total):
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(st):
                for off in range(0, total, piece):
                    dst[off:off + piece].copy_(pin[off:off + piece], non_blocking=True)
            t1 = time.perf_counter()
            st.synchronize()
            t2 = time.perf_counter()
        print("H2D piece %8d B: enqueue %.2f ms, done %.2f ms -> %.1f GB/s" % (piece, (t1 - t0) * 1e3, (t2 - t0) * 1e3, total / (t2 - t0) / 1e9))

if "stage" in what:
    print("staging report", ingest.staging_report())
    rng = np.random.default_rng(0)
    trajs = [rng.integers(0, 256, (128, 84, 84, 4), dtype=np.uint8) for _ in range(32)]
    nb = trajs[0].nbytes
    pin = torch.empty((32 * nb,), dtype=torch.uint8, pin_memory=True)
    dst = torch.empty((32 * nb,), dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream()
    for threads in (-1, 2, 8):
        for ship in (1 << 20, 2 << 20, 4 << 20):
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i, tr in enumerate(trajs):
                    L.check(h.xt_stage_rows(ctypes.c_void_p(pin.data_ptr() + i * nb), ctypes.c_void_p(tr.ctypes.data), nb,
                                            ctypes.c_void_p(dst.data_ptr() + i * nb), 0, ship, threads,
                                            ctypes.c_void_p(st.cuda_stream)), "stage")
                t1 = time.perf_counter()
                st.synchronize()
                t2 = time.perf_counter()
            print("threads %2d ship %8d: staged+enqueued %.2f ms, landed %.2f ms" % (threads, ship, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
    assert np.array_equal(dst.cpu().numpy().reshape(32, -1)[5], trajs[5].reshape(-1))

import bench  # noqa: E402


def profile(fn, n, top=22):
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n):
        fn(i)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(top)


if "ppo" in what:
    from xingtian_amd.algorithm import alg_builder
    from oracle import returns
    model_info = {"actor": {"model_name": "PpoCnn", "state_dim": list(bench.STATE_DIM), "action_dim": bench.A_DIM,
                            "input_dtype": "uint8",
                            "model_config": dict(bench.CFG, SUMMARY=False, VF_SHARE_LAYERS=True, activation="relu",
                                                 hidden_sizes=list(bench.HIDDEN), action_type="Categorical", SEED=0)}}
    alg = alg_builder("PPO", model_info, {"instance_num": 32, "agent_num": 1})
    obs, action, logp, value, reward, done = bench.synth_rollout(132, 32)
    trajs = []
    for i in range(32):
        a, ov, tg = returns.gae(value[i].reshape(-1, 1), reward[i].copy(), done[i])
        sl = slice(i * 128, (i + 1) * 128)
        trajs.append({"cur_state": obs[sl], "action": action[sl], "logp": logp[sl].reshape(-1, 1), "adv": a,
                      "old_value": ov, "target_value": tg})
    tm = {"prep": 0.0, "train": 0.0, "w": 0.0}

    def one(i):
        t0 = time.perf_counter()
        for tr in trajs:
            alg.prepare_data(tr)
        t1 = time.perf_counter()
        alg.train(episode_num=i)
        t2 = time.perf_counter()
        alg.get_weights()
        t3 = time.perf_counter()
        tm["prep"] += t1 - t0; tm["train"] += t2 - t1; tm["w"] += t3 - t2

    for i in range(5):
        one(i)
    tm = {k: 0.0 for k in tm}
    for i in range(20):
        one(i)
    print("PPO plugin path: prepare %.3f ms, train %.3f ms, get_weights %.3f ms" % tuple(1e3 * tm[k] / 20 for k in ("prep", "train", "w")))
    profile(one, 10)

if "impala" in what:
    from xingtian_amd.algorithm import alg_builder
    w = bench.IMPALA["breakout_impala"]
    data = bench.synth_impala(7, 128 * 8, 84, 4)
    model_info = {"actor": {"model_name": "ImpalaCnnOpt", "state_dim": [84, 84, 4], "input_dtype": "uint8", "state_mean": 0.0,
                            "state_std": 255.0, "action_dim": 4,
                            "model_config": {"LR": w["lr"], "sample_batch_step": 128, "grad_norm_clip": 40.0, "SEED": 0}}}
    alg = alg_builder("IMPALAOpt", model_info, {"instance_num": 32, "agent_num": 1, "prepare_times_per_train": 1, "BATCH_SIZE": 512})
    msgs = []
    for i in range(8):
        sl = slice(i * 128, (i + 1) * 128)
        msgs.append({"cur_state": data["obs"][sl], "logit": data["logit"][sl], "action": data["action"][sl],
                     "done": list(data["done"][sl]), "reward": list(data["reward"][sl])})

    def one_i(i):
        alg.prepare_data(msgs[i % 8])
        alg.train(episode_num=i)
        alg.get_weights()

    for i in range(20):
        one_i(i)
    t0 = time.perf_counter()
    for i in range(200):
        one_i(i)
    print("IMPALA plugin path: %.3f ms per train" % ((time.perf_counter() - t0) * 5))
    profile(one_i, 200)
