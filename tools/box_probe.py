#!/usr/bin/env python
"""What kind of box is this?  A quick PPO update timing, launch-to-launch periods of repeated vs alternating small kernels,
and -- when the update is slow -- everything rocm-smi says about the GPU.  One box in eight of the pool ran the same build
1.5x slower INSIDE the replayed graph with normal isolated kernel timings (profiles/r03_bench_slowbox.json); this collects
evidence about why.  GPU box; prints one JSON line."""
import ctypes
import json
import os
import subprocess
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

dev = torch.device("cuda", 0)
lib = L.load()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
obs, action, logp, value, reward, done = bench.synth_rollout(0)
n = obs.shape[0]
net = HipActorCritic(netspec.ppo_cnn(bench.STATE_DIM, bench.A_DIM, bench.HIDDEN, "relu", True), max_batch=320, seed=0)
args = (d(obs), d(np.stack([np.random.default_rng(4 + i).permutation(n) for i in range(4)]).astype(np.int32)), d(action), d(logp),
        d(np.random.default_rng(1).standard_normal(n)), d(np.random.default_rng(3).standard_normal(n).astype(np.float32)),
        d(np.random.default_rng(2).standard_normal(n)))
cfg = net.make_ppo_cfg(bench.CFG)


def update_ms(graph, reps=10):
    for _ in range(3):
        net.ppo_train(cfg, *args, use_graph=graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        net.ppo_train(cfg, *args, use_graph=graph)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


out = {"update_ms_graph": update_ms(True), "update_ms_eager": update_ms(False, 4)}
# launch-to-launch period of tiny kernels: the same one repeated vs four different ones in turn
state = torch.zeros(8, dtype=torch.float32, device=dev)
buf = torch.zeros(4096, dtype=torch.float32, device=dev)
scratch = torch.zeros(2048, dtype=torch.float32, device=dev)
u8 = torch.zeros(4096, dtype=torch.uint8, device=dev)
u8b = torch.zeros(8192, dtype=torch.uint8, device=dev)
dbl = torch.zeros(512, dtype=torch.float64, device=dev)
st = L.stream_ptr()
A = lambda: lib.xt_adam_state_init(L.ptr(state), st)
B = lambda: lib.xt_pad_channels(L.ptr(u8), L.ptr(u8b), 1024, 3, 4, 1, 0, st)
C = lambda: lib.xt_adv_normalize_f64(L.ptr(dbl), 512, 1e-8, None, st)
D = lambda: lib.xt_heads_fwd(L.ptr(buf), L.ptr(buf), 4, 256, 4, L.ptr(buf), L.ptr(buf), L.ptr(buf), L.ptr(buf), L.ptr(scratch), L.ptr(scratch), st)


def period_us(seq, reps=400):
    for f in seq:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        seq[i % len(seq)]()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


out["tiny_same_us"] = period_us([A])
out["tiny_alternating_us"] = period_us([A, B, C, D])
# per-layer kernels: isolated back-to-back period (what bench.py calls isolated)
idx = args[1][0, :320].contiguous()
out["conv2_bwd_isolated_us"] = 1e3 * net.time_layer(1, 3, args[0], idx, 320, 50)
out["conv1_fwd_isolated_us"] = 1e3 * net.time_layer(0, 0, args[0], idx, 320, 50)
slow = out["update_ms_graph"] > 8.5
out["slow_box"] = slow
if slow or "--dump" in sys.argv:
    for name, cmd in (("smi_all", ["rocm-smi", "--showall"]), ("smi_fw", ["rocm-smi", "--showfwinfo"]),
                      ("smi_bus", ["rocm-smi", "--showbus", "--showpids", "--showmemuse", "--showvoltage", "--showtemp"])):
        try:
            out[name] = subprocess.run(cmd, capture_output=True, text=True, timeout=20).stdout[-6000:]
        except Exception as e:      # noqa: BLE001
            out[name] = repr(e)
print(json.dumps(out))
