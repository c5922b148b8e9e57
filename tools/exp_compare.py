#!/usr/bin/env python
"""Experiment helper: one B = 320 PpoCnn minibatch gradient (+ a 1000-frame ImpalaCnnOpt one) with the library given on the
command line -> .npy; with two files, print the per-tensor relative L2 difference.  `python tools/exp_compare.py dump <lib.so> <out.npz>`
/ `python tools/exp_compare.py diff a.npz b.npz`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dump(lib_path, out):
    import torch
    from xingtian_amd import lib as L
    L.LIB_PATH = os.path.abspath(lib_path)
    if len(sys.argv) > 4:           # optional xt_tuning knobs as JSON
        import json
        L.set_tuning(**json.loads(sys.argv[4]))
    import bench
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rng = np.random.default_rng(3)
    b = 320
    spec = netspec.ppo_cnn((84, 84, 4), 4, (256,), "relu", True)
    net = HipActorCritic(spec, max_batch=b, seed=1)
    obs = rng.integers(0, 256, (b, 84, 84, 4)).astype(np.uint8)
    act = rng.integers(0, 4, b).astype(np.int32)
    logp = (-np.abs(rng.standard_normal(b)) - 0.5).astype(np.float32)
    adv, oldv = rng.standard_normal(b), rng.standard_normal(b).astype(np.float32)
    tgt = oldv + rng.standard_normal(b)
    net.ppo_step(net.make_ppo_cfg(bench.CFG), d(obs), None, d(act), d(logp), d(adv), d(oldv), d(tgt), apply=False)
    torch.cuda.synchronize()
    res = {"ppo/" + k: v for k, v in net.grads_dict().items()}
    w = bench.IMPALA["pong_impala_speedup"]
    data = bench.synth_impala(7, 1000, w["dim"], w["a_dim"])
    ispec = netspec.impala_cnn_opt((42, 42, 4), 6, 128.0, 128.0, "uint8")
    inet = HipActorCritic(ispec, max_batch=1000, seed=2)
    inet.impala_step(inet.make_impala_cfg(1e-3, 40.0, 50), d(data["obs"]), d(data["logit"]), d(data["action"]),
                     d(data["done"].astype(np.uint8)), d(data["reward"].astype(np.float32)), apply=False)
    torch.cuda.synchronize()
    res.update({"impala/" + k: v for k, v in inet.grads_dict().items()})
    np.savez(out, **res)


def diff(a, b):
    x, y = np.load(a), np.load(b)
    worst = 0.0
    for k in x.files:
        e = float(np.linalg.norm(x[k] - y[k]) / (np.linalg.norm(x[k]) + 1e-30))
        worst = max(worst, e)
        print("%-48s rel L2 diff %.3e" % (k, e))
    print("worst", worst)


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3])
    else:
        diff(sys.argv[2], sys.argv[3])
