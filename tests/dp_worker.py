"""Rank program of the multi-process data-parallel GPU tests (launched by tests/test_gpu_dp.py through
torch.distributed.run).  Every rank builds the same HipActorCritic replica on the ONE visible GPU and runs the
PRODUCT data-parallel path (xingtian_amd.parallel.dp_ppo_update / dp_impala_step); the gradient all-reduce goes through
``gloo`` (RCCL refuses two ranks on one device; gloo accepts device tensors), so everything but the transport is what
``bench.py --gpus N`` runs.  Rank r writes its final parameters to <outdir>/params_<mode>_r<r>.npy."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def ppo_case():
    """shared by the worker and the single-process reference in the test"""
    from xingtian_amd.model import netspec
    spec = netspec.ppo_cnn((42, 42, 4), 4, (64,), "relu", True)
    cfg = dict(LR=2.5e-4, LOSS_CLIPPING=0.1, ENTROPY_LOSS=0.003, VF_CLIP=5.0, CRITIC_LOSS_COEF=1.0,
               MAX_GRAD_NORM=5.0, BATCH_SIZE=32, NUM_SGD_ITER=2)
    return spec, cfg, 80        # 80 rows: 32 + 32 + 16 (short last minibatch: 8 rows per rank)


def ppo_rollout(seed, n):
    from test_gpu_learner import synth_ppo_rollout
    rng = np.random.default_rng(seed)
    obs, lab = synth_ppo_rollout(rng, n, (42, 42, 4), 4)
    perms = np.stack([rng.permutation(n) for _ in range(2)]).astype(np.int32)
    return obs, lab, perms


def impala_case():
    from xingtian_amd.model import netspec
    spec = netspec.impala_cnn_opt((42, 42, 4), 6, 128.0, 128.0)
    rng = np.random.default_rng(77)
    tlen, ntraj = 10, 5          # 5 trajectories over 2 ranks: shards of 3 and 2
    n = tlen * ntraj
    data = dict(obs=rng.integers(0, 256, (n, 42, 42, 4)).astype(np.uint8),
                bp=rng.standard_normal((n, 6)).astype(np.float32), act=rng.integers(0, 6, n).astype(np.int32),
                done=(rng.random(n) < 0.1).astype(np.uint8), rew=rng.choice([-1.0, 0.0, 1.0], n).astype(np.float32))
    return spec, data, tlen, ntraj


def main():
    outdir, mode = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    if os.environ.get("XT_TEST_LIB"):      # tests/test_gpu_wt_stress.py: the plain-store twin of the library
        from xingtian_amd import lib as _lib
        _lib.LIB_PATH = os.path.join(ROOT, "xingtian_amd", os.environ["XT_TEST_LIB"])
    from xingtian_amd import parallel
    from xingtian_amd.model.hip_net import HipActorCritic
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if mode in ("strict", "weak"):
        spec, cfg, n = ppo_case()
        net = HipActorCritic(spec, max_batch=cfg["BATCH_SIZE"], seed=5)
        parallel.broadcast_weights_(net.params)
        obs, lab, perms = ppo_rollout(100 if mode == "strict" else 200 + rank, n)
        if mode == "weak":
            _, _, perms = ppo_rollout(100, n)          # the SAME permutation of local row numbers on every rank
        steps = parallel.dp_ppo_update(net, cfg, net.to_device_obs(obs), d(perms), d(lab[0]), d(lab[1].reshape(-1)),
                                       d(lab[2].reshape(-1)), d(lab[3].reshape(-1)), d(lab[4].reshape(-1)), rank, world,
                                       mode=mode)
        assert steps == 6
    elif mode in ("hook", "hook_overlap"):
        # weak-mode update through the library's gradient-exchange hook (one C call, eager enqueue): one bucket, or two
        # with the first exchanged on the side stream right after the first backward launch
        spec, cfg, n = ppo_case()
        net = HipActorCritic(spec, max_batch=cfg["BATCH_SIZE"], seed=5)
        parallel.broadcast_weights_(net.params)
        obs, lab, _ = ppo_rollout(200 + rank, n)
        _, _, perms = ppo_rollout(100, n)
        ex = parallel.TorchDistExchange(net)
        ex.record_calls = True
        ex.attach(overlap=(mode == "hook_overlap"))
        c = net.make_ppo_cfg(cfg, grad_scale=1.0 / world, global_batch=0)
        net.ppo_train(c, net.to_device_obs(obs), d(perms), d(lab[0]), d(lab[1].reshape(-1)), d(lab[2].reshape(-1)),
                      d(lab[3].reshape(-1)), d(lab[4].reshape(-1)), use_graph=False)
        torch.cuda.synchronize()
        ex.detach()
        nflat = net.params.numel()
        off_a = spec.layers[-1].param_off
        want = [(0, nflat)] * 6 if mode == "hook" else [(off_a, nflat - off_a), (0, off_a)] * 6
        assert ex.calls == want, (ex.calls[:4], want[:4])
    elif mode == "strict_hook":
        # strict sharding INSIDE xt_net_ppo_train (xt_ppo_cfg.shard_rank / shard_world, ABI 9): every rank passes the same
        # rollout and permutations, the library takes this rank's rows of every global minibatch and exchanges through the hook
        spec, cfg, n = ppo_case()
        net = HipActorCritic(spec, max_batch=cfg["BATCH_SIZE"], seed=5)
        parallel.broadcast_weights_(net.params)
        obs, lab, perms = ppo_rollout(100, n)
        ex = parallel.TorchDistExchange(net)
        ex.record_calls = True
        ex.attach()
        c = net.make_ppo_cfg(cfg, grad_scale=1.0, global_batch=0, shard_rank=rank, shard_world=world)
        net.ppo_train(c, net.to_device_obs(obs), d(perms), d(lab[0]), d(lab[1].reshape(-1)), d(lab[2].reshape(-1)),
                      d(lab[3].reshape(-1)), d(lab[4].reshape(-1)), use_graph=False)
        torch.cuda.synchronize()
        ex.detach()
        assert len(ex.calls) == 6
    elif mode in ("impala_rms", "impala_sched"):
        # data-parallel IMPALA with opt_type rmsprop / an lr_schedule step size: through the exchange hook of
        # xt_net_impala_train, which applies the configured optimiser to the exchanged gradient itself
        spec, data, tlen, ntraj = impala_case()
        net = HipActorCritic(spec, max_batch=tlen * ntraj, seed=5)
        parallel.broadcast_weights_(net.params)
        if mode == "impala_rms":
            net.set_optimizer("rmsprop")
        c = net.make_impala_cfg(1e-3, 40.0, tlen, opt_type="rmsprop" if mode == "impala_rms" else "adam")
        for step in range(2):
            lr_steps = d(np.asarray([7e-4 / (step + 1)], np.float32)) if mode == "impala_sched" else None
            parallel.dp_impala_step(net, c, 1e-3, 40.0, d(data["obs"]), d(data["bp"]), d(data["act"]), d(data["done"]),
                                    d(data["rew"]), ntraj, tlen, rank, world, lr_steps=lr_steps)
    elif mode in ("impala_one_traj", "impala_rms_one_traj"):
        # ONE trajectory over two ranks: rank 1's shard is empty and contributes a zero gradient -- through the step-wise
        # branch (Adam) and through the hook branch (rmsprop), which must accept the same inputs
        spec, data, tlen, _ = impala_case()
        net = HipActorCritic(spec, max_batch=tlen, seed=5)
        parallel.broadcast_weights_(net.params)
        rms = mode == "impala_rms_one_traj"
        if rms:
            net.set_optimizer("rmsprop")
        c = net.make_impala_cfg(1e-3, 40.0, tlen, opt_type="rmsprop" if rms else "adam")
        sl = slice(0, tlen)
        for _ in range(2):
            parallel.dp_impala_step(net, c, 1e-3, 40.0, d(data["obs"][sl]), d(data["bp"][sl]), d(data["act"][sl]),
                                    d(data["done"][sl]), d(data["rew"][sl]), 1, tlen, rank, world)
    elif mode == "impala":
        spec, data, tlen, ntraj = impala_case()
        net = HipActorCritic(spec, max_batch=tlen * ntraj, seed=5)
        parallel.broadcast_weights_(net.params)
        c = net.make_impala_cfg(1e-3, 40.0, tlen)
        for _ in range(2):
            parallel.dp_impala_step(net, c, 1e-3, 40.0, d(data["obs"]), d(data["bp"]), d(data["act"]), d(data["done"]),
                                    d(data["rew"]), ntraj, tlen, rank, world)
    else:
        raise SystemExit("unknown mode " + mode)
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, "params_{}_r{}.npy".format(mode, rank)), net.params.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
