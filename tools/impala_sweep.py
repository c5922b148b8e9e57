#!/usr/bin/env python
"""A/B sweep of xt_tuning knobs on the two IMPALA workloads (HBM-resident us per train); GPU box."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xingtian_amd import lib as L  # noqa: E402
from xingtian_amd.model import netspec  # noqa: E402
from xingtian_amd.model.hip_net import HipActorCritic  # noqa: E402

for a in sys.argv[1:]:          # python tools/impala_sweep.py path/to/lib.so [json sweep]: A/B of two builds
    if a.endswith(".so"):
        L.LIB_PATH = os.path.abspath(a)
        sys.argv.remove(a)
        break
dev = torch.device("cuda", 0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def run(key, knobs):
    old = L.set_tuning(**knobs)
    try:
        w = bench.IMPALA[key]
        f, trains = w["frames_per_train"], w["trains"]
        data = bench.synth_impala(7, f * trains, w["dim"], w["a_dim"])
        spec = netspec.impala_cnn_opt((w["dim"], w["dim"], 4), w["a_dim"], w["mean"], w["std"], "uint8")
        net = HipActorCritic(spec, max_batch=f, seed=0)
        cfg = net.make_impala_cfg(w["lr"], 40.0, w["t_len"])
        args = (d(data["obs"]), f, d(data["logit"]), d(data["action"]), d(data["done"].astype(np.uint8)),
                d(data["reward"].astype(np.float32)))
        for _ in range(3):
            net.impala_train(cfg, *args, use_graph=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            net.impala_train(cfg, *args, use_graph=True)
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / (10 * trains)
    finally:
        L.set_tuning(**old)


SWEEP = [{}, {"wgrad_split_target": 256}, {"wgrad_split_target": 1024}, {"wgrad_split_target": 2048},
         {"fwd_split_target": 128}, {"fwd_split_target": 512}, {"direct": 0}, {"direct_all": 1}, {"fwd_two_groups": 0},
         {"direct_waves": 768}, {"direct_waves": 3072}, {"reduce_z_lanes": 4}, {"reduce_z_lanes": 16}, {"defer_splitk": 0}]
if len(sys.argv) > 1:          # python tools/impala_sweep.py '[{}, {"fwd_xcd_chunk": 0}]'
    import json
    SWEEP = json.loads(sys.argv[1])
for key in ("breakout_impala", "pong_impala_speedup"):
    for knobs in SWEEP:
        print("%-22s %-32s %8.1f us/train" % (key, knobs, run(key, knobs)), flush=True)
