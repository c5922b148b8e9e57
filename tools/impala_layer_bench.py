#!/usr/bin/env python
"""Per-kernel timing of the ImpalaCnnOpt layers (HIP events via xt_net_time_layer): forward, weight gradient, input
gradient, and the fused backward launch, for the two IMPALA workloads of bench.py.  GPU box."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xingtian_amd.model import netspec
from xingtian_amd.model.hip_net import HipActorCritic
from xingtian_amd import lib as L

if len(sys.argv) > 1:          # python tools/impala_layer_bench.py '{"direct_fwd": 0}'
    import json
    L.set_tuning(**json.loads(sys.argv[1]))

for dim, a_dim, B, mean, std in ((84, 4, 128, 0.0, 255.0), (42, 6, 1000, 128.0, 128.0)):
    spec = netspec.impala_cnn_opt((dim, dim, 4), a_dim, mean, std)
    net = HipActorCritic(spec, max_batch=B, seed=0)
    rng = np.random.default_rng(0)
    obs = torch.from_numpy(rng.integers(0, 256, (B, dim, dim, 4), dtype=np.uint8)).cuda()
    net.forward(obs)
    print("---- %dx%d B=%d" % (dim, dim, B))
    for li, lay in enumerate(spec.layers):
        flops = 2.0 * B * lay.OH * lay.OW * lay.N * lay.K
        for which, nm in ((0, "fwd"), (1, "wgrad")) + (((2, "dgrad"), (3, "fused")) if li else ()):
            for rep in range(2):
                ms = net.time_layer(li, which, obs, None, B, reps=30)
            print("%-26s %-5s %8.2f us  %6.1f TFLOP/s" % (lay.name, nm, ms * 1e3, (2 * flops if which == 3 else flops) / ms / 1e9))
