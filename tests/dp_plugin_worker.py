"""Rank program of tests/test_gpu_dp_plugin.py: N learner PROCESSES on the one visible GPU, each building the SAME
Algorithm / Model pair through ``alg_builder`` exactly as xt/framework/learner.py:518-525 does; ``WORLD_SIZE > 1`` (set by
torch.distributed.run) makes the model one data-parallel replica (xingtian_amd/parallel.py::LearnerDP).  The gradient
exchange is gloo from the library's host hook (``DP_EXCHANGE: torch``; RCCL refuses two ranks on one device) or the direct
all-reduce over hipIpc-mapped memory captured into the update's hipGraph (``DP_EXCHANGE: direct``).
Rank r writes <outdir>/<case>_r<r>.npz = final parameters + the losses train() returned + its publisher answers."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

T_LEN, N_TRAJ, A_DIM, DIM = 16, 6, 4, 42
PPO_MC = dict(LR=2.5e-4, LOSS_CLIPPING=0.1, ENTROPY_LOSS=0.003, VF_CLIP=5.0, CRITIC_LOSS_COEF=1.0, MAX_GRAD_NORM=5.0,
              BATCH_SIZE=32, NUM_SGD_ITER=2, VF_SHARE_LAYERS=True, activation="relu", hidden_sizes=[64],
              action_type="Categorical", SEED=3, SUMMARY=False)
UPDATES = int(os.environ.get("XT_DP_UPDATES", "2"))       # (the soak test raises it)


def ppo_model_info(extra):
    return {"actor": {"model_name": "PpoCnn", "state_dim": [DIM, DIM, 4], "action_dim": A_DIM, "input_dtype": "uint8",
                      "type": "learner", "model_config": dict(PPO_MC, **extra)}}


def ppo_trajs(update):
    """N_TRAJ trajectories of update `update` (the same on every rank)"""
    from test_gpu_learner import synth_ppo_rollout
    out = []
    for i in range(N_TRAJ):
        rng = np.random.default_rng(5000 + 100 * update + i)
        obs, lab = synth_ppo_rollout(rng, T_LEN, (DIM, DIM, 4), A_DIM)
        out.append({"cur_state": obs, "action": lab[0], "logp": lab[1], "adv": lab[2], "old_value": lab[3],
                    "target_value": lab[4]})
    return out


IMPALA_T, IMPALA_MSGS, IMPALA_ENVS = 10, 4, 2          # 4 messages x 2 envs x T=10 -> 80 frames per train, chunks of 40


def impala_model_info(extra):
    return {"actor": {"model_name": "ImpalaCnnOpt", "state_dim": [DIM, DIM, 4], "input_dtype": "uint8", "type": "learner",
                      "state_mean": 128.0, "state_std": 128.0, "action_dim": 6,
                      "model_config": dict({"LR": 1e-3, "sample_batch_step": IMPALA_T, "grad_norm_clip": 40.0, "SEED": 4},
                                           **extra)}}


IMPALA_ALG = {"instance_num": 4, "agent_num": 1, "prepare_times_per_train": IMPALA_MSGS, "BATCH_SIZE": 40}


def impala_msgs(update):
    out = []
    for i in range(IMPALA_MSGS):
        rng = np.random.default_rng(7000 + 100 * update + i)
        n = IMPALA_ENVS * IMPALA_T
        out.append({"cur_state": rng.integers(0, 256, (n, DIM, DIM, 4)).astype(np.uint8),
                    "logit": rng.standard_normal((n, 6)).astype(np.float32), "action": rng.integers(0, 6, n).astype(np.int32),
                    "done": list(rng.random(n) < 0.1), "reward": list(rng.choice([-1.0, 0.0, 1.0], n))})
    return out


def yaml_config(extra):
    """examples/breakout_ppo.yaml as the reference ships it (parsed, from the reference-executed fixture
    tests/golden/learner_config.json) + the data-parallel keys in model_config: what INTEGRATION.md section 3b tells a
    maintainer to hand to build_learner_algorithm in every rank"""
    import copy
    import json
    with open(os.path.join(ROOT, "tests", "golden", "learner_config.json")) as f:
        cfg = copy.deepcopy(json.load(f)["examples/breakout_ppo.yaml"]["config"])
    cfg["model_para"]["actor"]["model_config"].update(dict(extra, SEED=6))
    return cfg


def yaml_trajs(update, env_num):
    from test_gpu_learner import synth_ppo_rollout
    out = []
    for i in range(env_num):
        rng = np.random.default_rng(9000 + 100 * update + i)
        obs, lab = synth_ppo_rollout(rng, 128, (84, 84, 4), 4)
        out.append({"cur_state": obs, "action": lab[0], "logp": lab[1], "adv": lab[2], "old_value": lab[3],
                    "target_value": lab[4]})
    return out


# ---- BASELINE configs[3] / configs[4] at their PER-RANK shapes: examples/breakout_ppo.yaml
# (PpoCnn 84x84x4, BATCH_SIZE 320 -> 40-row shards at 8 ranks) and examples/pong_impala_speedup.yaml (ImpalaCnnOpt
# 42x42x4, A = 6, 4 messages x 5 envs x T = 50 = 20 trajectories per 1000-frame chunk -> 3,3,3,3,2,2,2,2 at 8 ranks)
C3_SEED, C4_SEED = 6, 8


def c3_config(extra, one_step=False):
    cfg = yaml_config(extra)
    if one_step:        # exactly ONE SGD step per update: the exchanged gradient of that step can be read back
        cfg["model_para"]["actor"]["model_config"]["NUM_SGD_ITER"] = 1
    return cfg


def c3_trajs(update, n_traj, t_len):
    """``n_traj`` trajectories of ``t_len`` rows (7 x 128 = 896 rows = 2 minibatches of 320 + one of 256 per epoch)"""
    from test_gpu_learner import synth_ppo_rollout
    out = []
    for i in range(n_traj):
        rng = np.random.default_rng(11000 + 100 * update + i)
        obs, lab = synth_ppo_rollout(rng, t_len, (84, 84, 4), 4)
        out.append({"cur_state": obs, "action": lab[0], "logp": lab[1], "adv": lab[2], "old_value": lab[3],
                    "target_value": lab[4]})
    return out


def c3_shape(feed, world, one_step):
    """(trajectories, rows per trajectory) of one update"""
    if one_step:
        return (5, 64) if feed == "replicated" else (world, 320 // world)       # 320 rows: one global minibatch
    return (7, 128) if feed == "replicated" else (world, 128)     # sharded feeds: one 128-row trajectory per rank


def c4_config(extra, batch_size=None, max_batch=None):
    import copy
    import json
    with open(os.path.join(ROOT, "tests", "golden", "learner_config.json")) as f:
        cfg = copy.deepcopy(json.load(f)["examples/pong_impala_speedup.yaml"]["config"])
    cfg["model_para"]["actor"]["model_config"].update(dict(extra, SEED=C4_SEED))
    if batch_size:
        cfg["alg_para"]["alg_config"]["BATCH_SIZE"] = int(batch_size)
    if max_batch:
        cfg["model_para"]["actor"]["model_config"]["MAX_BATCH"] = int(max_batch)
    return cfg


def c4_msg(update, k, envs=5, t_len=50):
    """message k of update `update`: `envs` trajectories of T = 50, env-major rows (atari_impala_opt.py:96-109)"""
    rng = np.random.default_rng(13000 + 100 * update + k)
    n = envs * t_len
    return {"cur_state": rng.integers(0, 256, (n, 42, 42, 4)).astype(np.uint8),
            "logit": rng.standard_normal((n, 6)).astype(np.float32), "action": rng.integers(0, 6, n).astype(np.int32),
            "done": list(rng.random(n) < 0.02), "reward": list(rng.choice([-1.0, 0.0, 1.0], n, p=[0.05, 0.9, 0.05]))}


def exchanged_gradient(alg):
    """the gradient the LAST SGD step applied (after the exchange), as a host array"""
    dp, net = alg.dp, alg.actor.net
    if dp is not None and hasattr(dp, "exchanged_gradient"):
        return dp.exchanged_gradient(net)
    import torch
    torch.cuda.synchronize()
    return net.grads.detach().cpu().numpy().copy()


def main():
    outdir, case = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    from xingtian_amd.algorithm import alg_builder
    alg_kind, mode, feed, exchange = case.split("-")
    extra = {"DP": mode, "DP_FEED": feed, "DP_EXCHANGE": exchange, "DP_BACKEND": "gloo", "DP_DEVICE": 0}
    losses, answers = [], []
    grad1 = None
    if alg_kind in ("c3", "c3g"):
        from xingtian_amd.config import build_learner_algorithm
        one = alg_kind == "c3g"
        alg = build_learner_algorithm(c3_config(extra, one_step=one))
        n_traj, t_len = c3_shape(feed, world, one)
        for u in range(UPDATES):
            for k, tr in enumerate(c3_trajs(u, n_traj, t_len)):
                if feed == "sharded" and k % world != rank:
                    continue
                alg.prepare_data(tr)
            losses.append(float(alg.train(episode_num=u)))
            answers.append(bool(alg.checkpoint_ready(u)))
            if one and u == 0:
                grad1 = exchanged_gradient(alg)
    elif alg_kind == "c4":
        from xingtian_amd.config import build_learner_algorithm
        if mode == "weak":        # every rank trains a full 250-frame chunk of its OWN message: global chunk world x 250
            cfg = c4_config(extra, batch_size=250)
            cfg["alg_para"]["alg_config"]["prepare_times_per_train"] = 1
        else:
            cfg = c4_config(extra)
        alg = build_learner_algorithm(cfg)
        for u in range(UPDATES):
            if mode == "weak":
                alg.prepare_data(c4_msg(u, rank))
            else:
                for k in range(4):
                    if feed == "sharded" and k % world != rank:
                        continue
                    alg.prepare_data(c4_msg(u, k))
            losses.append(float(alg.train(episode_num=u)))
            answers.append(bool(alg.checkpoint_ready(u)))
            if u == 0:
                grad1 = exchanged_gradient(alg)     # (one 1000-frame chunk per train = one step)
    elif alg_kind == "yaml":
        from xingtian_amd.config import build_learner_algorithm
        cfg = yaml_config(extra)
        alg = build_learner_algorithm(cfg)
        assert alg.prepare_data_times == cfg["env_num"] == 10
        for u in range(UPDATES):
            for tr in yaml_trajs(u, cfg["env_num"]):
                alg.prepare_data(tr)
            losses.append(float(alg.train(episode_num=u)))
            answers.append(bool(alg.checkpoint_ready(u)))
    elif alg_kind == "ppo":
        alg = alg_builder("PPO", ppo_model_info(extra), {"instance_num": N_TRAJ, "agent_num": 1})
        for u in range(UPDATES):
            for k, tr in enumerate(ppo_trajs(u)):
                if feed == "sharded" and k % world != rank:
                    continue                       # "sharded": this rank is only ever handed its own trajectories
                alg.prepare_data(tr)
            losses.append(float(alg.train(episode_num=u)))
            answers.append(bool(alg.checkpoint_ready(u)))
    else:
        alg = alg_builder("IMPALAOpt", impala_model_info(extra), dict(IMPALA_ALG))
        for u in range(UPDATES):
            for k, m in enumerate(impala_msgs(u)):
                if feed == "sharded" and k % world != rank:
                    continue
                alg.prepare_data(m)
            losses.append(float(alg.train(episode_num=u)))
            answers.append(bool(alg.checkpoint_ready(u)))
    torch.cuda.synchronize()
    dp = alg.dp
    assert dp is not None and dp.world == world and dp.rank == rank
    assert alg.actor.use_graph == (exchange != "torch")
    st = dp.status() or {}
    info = dp.comm.info() if hasattr(dp.comm, "info") else {}
    np.savez(os.path.join(outdir, "{}_r{}.npz".format(case, rank)), params=alg.actor.net.params.cpu().numpy(),
             losses=np.asarray(losses), answers=np.asarray(answers), if_save=np.asarray([alg.if_save(0) is not False]),
             grad1=np.zeros(0, np.float32) if grad1 is None else grad1,
             error_bits=np.asarray([int(st.get("error_bits", 0))]),
             ranks_on_device=np.asarray([int(info.get("ranks_on_device", 0))]), block_cap=np.asarray([int(info.get("block_cap", 0))]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
