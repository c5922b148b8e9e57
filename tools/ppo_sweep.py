#!/usr/bin/env python
"""A/B sweep of xt_tuning knobs on the headline PPO workload (HBM-resident ms per update); GPU box."""
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

for a in sys.argv[1:]:          # python tools/ppo_sweep.py path/to/lib.so [json sweep]: A/B of two builds
    if a.endswith(".so"):
        L.LIB_PATH = os.path.abspath(a)
        sys.argv.remove(a)
        break
dev = torch.device("cuda", 0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
obs, action, logp, value, reward, done = bench.synth_rollout(0)
n = obs.shape[0]
dobs, dact, dlogp = d(obs), d(action), d(logp)
adv = d(np.random.default_rng(1).standard_normal(n))
tgt = d(np.random.default_rng(2).standard_normal(n))
oldv = d(np.random.default_rng(3).standard_normal(n).astype(np.float32))
perm = d(np.stack([np.random.default_rng(4 + i).permutation(n) for i in range(4)]).astype(np.int32))


def run(knobs):
    old = L.set_tuning(**knobs)
    try:
        spec = netspec.ppo_cnn(bench.STATE_DIM, bench.A_DIM, bench.HIDDEN, "relu", True)
        net = HipActorCritic(spec, max_batch=320, seed=0)
        cfg = net.make_ppo_cfg(bench.CFG)
        for _ in range(3):
            net.ppo_train(cfg, dobs, perm, dact, dlogp, adv, oldv, tgt, use_graph=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(15):
            net.ppo_train(cfg, dobs, perm, dact, dlogp, adv, oldv, tgt, use_graph=True)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / 15
    finally:
        L.set_tuning(**old)


SWEEP = [{}, {"reduce_z_lanes": 16}, {"reduce_z_lanes": 32}, {"reduce_z_lanes": 4}, {"wgrad_split_target": 384},
         {"wgrad_split_target": 640}, {"wgrad_split_target": 768}, {"bwd_fit_slots": 0}, {"bwd_fit_slots": 640},
         {"fwd_split_target": 192}, {"fwd_split_target": 320}, {"direct_waves": 1024}, {"direct_waves": 2048},
         {"direct_max_waves": 4}, {"dgrad_halo": 0}, {"fwd_two_groups": 0}, {"defer_splitk": 0}, {}]
if len(sys.argv) > 1:          # python tools/ppo_sweep.py '[{}, {"direct_fwd": 0}]'
    import json
    SWEEP = json.loads(sys.argv[1])
for knobs in SWEEP:
    print("%-32s %8.3f ms/update" % (knobs, run(knobs)), flush=True)
