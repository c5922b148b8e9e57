#!/usr/bin/env python
"""Host cost of replaying the update's hipGraph (572 kernel nodes): wall time of the xt_net_ppo_train call itself (returns
when the launch is enqueued) next to the update's GPU time.  A launch that takes longer than the kernels makes the update
host-bound on that box.  GPU box; honours the HIP runtime's own environment variables (e.g. DEBUG_CLR_GRAPH_PACKET_CAPTURE)."""
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xingtian_amd.model import netspec  # noqa: E402
from xingtian_amd.model.hip_net import HipActorCritic  # noqa: E402

print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|MHz' | head -4", shell=True, capture_output=True, text=True).stdout)
dev = torch.device("cuda", 0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
obs, action, logp, value, reward, done = bench.synth_rollout(0)
n = obs.shape[0]
dobs, dact, dlogp = d(obs), d(action), d(logp)
adv = d(np.random.default_rng(1).standard_normal(n))
tgt = d(np.random.default_rng(2).standard_normal(n))
oldv = d(np.random.default_rng(3).standard_normal(n).astype(np.float32))
perm = d(np.stack([np.random.default_rng(4 + i).permutation(n) for i in range(4)]).astype(np.int32))
spec = netspec.ppo_cnn(bench.STATE_DIM, bench.A_DIM, bench.HIDDEN, "relu", True)
net = HipActorCritic(spec, max_batch=320, seed=0)
cfg = net.make_ppo_cfg(bench.CFG)
for _ in range(5):
    net.ppo_train(cfg, dobs, perm, dact, dlogp, adv, oldv, tgt, use_graph=True)
torch.cuda.synchronize()
for rep in range(3):
    call, total = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net.ppo_train(cfg, dobs, perm, dact, dlogp, adv, oldv, tgt, use_graph=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        call.append(1e3 * (t1 - t0)); total.append(1e3 * (t2 - t0))
    print("launch call %.3f ms (min %.3f)   update incl. sync %.3f ms (min %.3f)" %
          (np.mean(call), np.min(call), np.mean(total), np.min(total)))
