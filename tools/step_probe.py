#!/usr/bin/env python
"""Eager (no hipGraph) learner updates of the three benchmark workloads: the target of the rocprofv3 --pmc passes
(tools/profile_round.sh), so that every kernel of a step -- layer kernels, heads, gradient reduction, clip + Adam --
is counted in the context it runs in.  Usage: python tools/step_probe.py [ppo] [breakout_impala] [pong_impala_speedup]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xingtian_amd import lib as L  # noqa: E402
from xingtian_amd.model import netspec  # noqa: E402
from xingtian_amd.model.hip_net import HipActorCritic  # noqa: E402

which = [a for a in sys.argv[1:] if not a.startswith("-")] or ["ppo", "breakout_impala", "pong_impala_speedup"]
dev = torch.device("cuda", 0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
if "ppo" in which:
    obs, action, logp, value, reward, done = bench.synth_rollout(0)
    n = obs.shape[0]
    spec = netspec.ppo_cnn(bench.STATE_DIM, bench.A_DIM, bench.HIDDEN, "relu", True)
    net = HipActorCritic(spec, max_batch=320, seed=0)
    adv = torch.empty((n,), dtype=torch.float64, device=dev)
    tgt = torch.empty((n,), dtype=torch.float64, device=dev)
    oldv = torch.empty((n,), dtype=torch.float32, device=dev)
    dv, dr, dd = d(value), d(reward), d(done.astype(np.uint8))
    L.check(net.lib.xt_gae_f64(L.ptr(dv), L.ptr(dr), L.ptr(dd), L.ptr(adv), L.ptr(tgt), L.ptr(oldv), 32, 128, 0.99, 0.95,
                               L.stream_ptr()), "gae")
    rng = np.random.default_rng(1)
    perm = d(np.stack([rng.permutation(n) for _ in range(4)]).astype(np.int32))
    cfg = net.make_ppo_cfg(bench.CFG)
    dobs, dact, dlogp = d(obs), d(action), d(logp)
    for _ in range(2):
        net.ppo_train(cfg, dobs, perm, dact, dlogp, adv, oldv, tgt, use_graph=False)
    torch.cuda.synchronize()
    print("ppo: 2 updates x 52 SGD steps done")
for key in ("breakout_impala", "pong_impala_speedup"):
    if key not in which:
        continue
    w = bench.IMPALA[key]
    f, trains = w["frames_per_train"], w["trains"]
    data = bench.synth_impala(7, f * trains, w["dim"], w["a_dim"])
    spec = netspec.impala_cnn_opt((w["dim"], w["dim"], 4), w["a_dim"], w["mean"], w["std"], "uint8")
    net = HipActorCritic(spec, max_batch=f, seed=0)
    cfg = net.make_impala_cfg(w["lr"], 40.0, w["t_len"])
    net.impala_train(cfg, d(data["obs"]), f, d(data["logit"]), d(data["action"]), d(data["done"].astype(np.uint8)),
                     d(data["reward"].astype(np.float32)), use_graph=False)
    torch.cuda.synchronize()
    print(key, ": %d trains done" % trains)
