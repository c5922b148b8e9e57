#!/usr/bin/env python
"""Runs the fused trunk forward (xt_net_time_layer which=4) in isolation: target for rocprofv3 --pmc passes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xingtian_amd.model import netspec  # noqa: E402
from xingtian_amd.model.hip_net import HipActorCritic  # noqa: E402

B = 320
spec = netspec.ppo_cnn((84, 84, 4), 4, (256,), "relu", True)
net = HipActorCritic(spec, max_batch=B, seed=0)
rng = np.random.default_rng(0)
obs = torch.from_numpy(rng.integers(0, 256, (1024, 84, 84, 4), dtype=np.uint8)).cuda()
idx = torch.from_numpy(rng.permutation(1024)[:B].astype(np.int32)).cuda()
which = int(sys.argv[1]) if len(sys.argv) > 1 else 4
print("ms per launch:", net.time_layer(0, which, obs, idx, B, reps=50))
