#!/usr/bin/env python
"""Diagnostic (GPU box): per-tensor gradient error of one ImpalaCnnOpt step against the float64 oracle, with the
bf16x3 first-layer kernels (default) and with the generic fp32 implicit-GEMM kernels (conv1_bf16x3 = 0)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nets
from xingtian_amd import lib as L
from xingtian_amd.model import netspec
from xingtian_amd.model.hip_net import HipActorCritic
from test_gpu_learner import oracle_params_for, rel_err

for dim, a_dim, tlen, ntraj, mean, std in ((84, 4, 128, 1, 0.0, 255.0), (42, 6, 50, 20, 128.0, 128.0)):
    for knobs in ({}, {"conv1_bf16x3": 0}):
        old = L.set_tuning(**knobs)
        spec = netspec.impala_cnn_opt((dim, dim, 4), a_dim, mean, std)
        ospec = nets.impala_cnn_opt_spec((dim, dim, 4), a_dim, mean, std)
        n = tlen * ntraj
        net = HipActorCritic(spec, max_batch=n, seed=0)
        params = oracle_params_for(net, ospec, seed=5)
        rng = np.random.default_rng(1)
        obs = rng.integers(0, 256, (n, dim, dim, 4)).astype(np.uint8)
        bp = rng.standard_normal((n, a_dim)).astype(np.float32)
        act = rng.integers(0, a_dim, n).astype(np.int32)
        done = rng.random(n) < 0.05
        rew = rng.choice([-2.0, 0.0, 1.0, 3.0], n).astype(np.float32)
        cfg = dict(LR=5e-4, grad_norm_clip=40.0, sample_batch_step=tlen, BATCH_SIZE=n)
        orc = nets.ImpalaLearnerOracle(ospec, params, cfg, np.float64)
        out = orc.step(obs, bp, act, done, rew, apply=True)
        c = net.make_impala_cfg(5e-4, 40.0, tlen)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        lo = net.impala_step(c, d(obs), d(bp), d(act), d(done.astype(np.uint8)), d(rew), apply=True)
        torch.cuda.synchronize()
        g = net.grads_dict()
        w = net.get_weights()
        print(dim, knobs, "loss", float(lo.cpu().numpy()[0]), out["loss"])
        for k, ref in out["grads"].items():
            init = np.asarray(params[k], np.float64).reshape(ref.shape)
            upd_ref = orc.net.params[k] - init
            upd_got = np.asarray(w[k], np.float64).reshape(ref.shape) - init
            print("   %-34s grad %.2e  update %.2e" % (k, rel_err(g[k].reshape(ref.shape), ref),
                                                     rel_err(upd_got, upd_ref) if np.linalg.norm(upd_ref) > 0 else 0.0))
        L.set_tuning(**old)
