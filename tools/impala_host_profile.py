#!/usr/bin/env python
"""cProfile of the IMPALAOpt plugin path (prepare_data -> train -> publish_weights) on the breakout_impala shape:
where the host time per 128-frame train goes.  GPU box."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xingtian_amd import transport  # noqa: E402
from xingtian_amd.algorithm import alg_builder  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "breakout_impala"
w = bench.IMPALA[key]
f = w["frames_per_train"]
data = bench.synth_impala(7, f * 8, w["dim"], w["a_dim"])
model_info = {"actor": {"model_name": "ImpalaCnnOpt", "state_dim": [w["dim"], w["dim"], 4], "input_dtype": "uint8",
                        "state_mean": w["mean"], "state_std": w["std"], "action_dim": w["a_dim"],
                        "model_config": {"LR": w["lr"], "sample_batch_step": w["t_len"], "grad_norm_clip": 40.0, "SEED": 0}}}
alg = alg_builder("IMPALAOpt", model_info, {"instance_num": 32, "agent_num": 1, "prepare_times_per_train": 1,
                                           "train_per_checkpoint": 1, "BATCH_SIZE": max(f, 512)})
msgs = []
for i in range(8):
    sl = slice(i * f, (i + 1) * f)
    msgs.append({"cur_state": data["obs"][sl], "logit": data["logit"][sl], "action": data["action"][sl],
                 "done": list(data["done"][sl]), "reward": list(data["reward"][sl])})
ring = transport.WeightsRing(slot_bytes=8 << 20, slots=3)
assert ring.pin()


def loop(n):
    for i in range(n):
        alg.prepare_data(msgs[i % 8])
        alg.train(episode_num=i)
        alg.publish_weights(ring)


loop(20)
t0 = time.perf_counter()
loop(300)
print("ms per train: %.3f" % (1e3 * (time.perf_counter() - t0) / 300))
pr = cProfile.Profile()
pr.enable()
loop(300)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
ring.close()
