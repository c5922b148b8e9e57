#!/usr/bin/env python
"""Per-kernel timing of the PpoCnn layers at B=320 (HIP events on the launch stream, via xt_net_time_layer).
Usage: python tools/layer_bench.py [path/to/libxt_mi355x.so]   -- prints us and TFLOP/s per kernel."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xingtian_amd import lib as L  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    L.LIB_PATH = os.path.abspath(sys.argv[1])
if os.environ.get("XT_KNOBS"):          # e.g. XT_KNOBS='{"wgrad_rows": 0}'
    import json
    L.set_tuning(**json.loads(os.environ["XT_KNOBS"]))
BS = [int(a) for a in sys.argv[1:] if a.isdigit()] or [320]
FUSED = "fused" in sys.argv[1:]      # time the fused dgrad+wgrad launch instead of the two separate kernels
from xingtian_amd.model import netspec  # noqa: E402
from xingtian_amd.model.hip_net import HipActorCritic  # noqa: E402

B = max(BS)
spec = netspec.ppo_cnn((84, 84, 4), 4, (256,), "relu", True)
net = HipActorCritic(spec, max_batch=B, seed=0)
rng = np.random.default_rng(0)
obs = torch.from_numpy(rng.integers(0, 256, (4096, 84, 84, 4), dtype=np.uint8)).cuda()
idx = torch.from_numpy(rng.permutation(4096)[:B].astype(np.int32)).cuda()
net.forward(obs[:B])      # fill activations with sane values
net.workspace.normal_(0, 0.1) if False else None
for B in BS:
    print("---- B =", B)
    for rep in range(2):
        tot = 0.0
        for li, lay in enumerate(spec.layers):
            flops = 2.0 * B * lay.OH * lay.OW * lay.N * lay.K
            kinds = ((0, "fwd"), (1, "wgrad"), (2, "dgrad")) if not FUSED else ((0, "fwd"), (3, "bwd") if li else (1, "wgrad"))
            for which, nm in kinds:
                if which == 2 and li == 0:
                    continue
                ms = net.time_layer(li, which, obs, idx, B, reps=50)
                tot += ms
                if rep == 1:
                    print("%-22s %-5s %8.2f us  %6.1f TFLOP/s" % (lay.name, nm, ms * 1e3, (2 * flops if which == 3 else flops) / ms / 1e9))
    print("sum of layer kernels: %.1f us (ideal at 157.3 TF: %.1f us)" % (tot * 1e3, 31.313e6 * B / 157.3e12 * 1e6))
