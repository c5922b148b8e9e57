"""Data parallelism THROUGH THE PLUGIN PAIR (north_star: "the learner shards minibatches data-parallel
across the GPUs" behind the registry): N learner processes each call ``alg_builder(...)`` as xt/framework/learner.py:518-525
does; under ``torchrun`` (WORLD_SIZE > 1) the model becomes one replica -- exchange attached, minibatches / chunks split,
weights published from rank 0.  Two ranks share the one GPU of the test box (gloo hook, or the direct all-reduce over
hipIpc-mapped memory inside the update's hipGraph).  Checked against the single-process update of the same plugin pair.
The reference's only analogue: xt/framework/trainer.py:32-136 (host-side gradient averaging, dead code)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(tmp_path, case, world=2, updates=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_plugin_worker.py"), str(tmp_path), case]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    if updates:
        env["XT_DP_UPDATES"] = str(updates)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert proc.returncode == 0, proc.stdout.decode()[-3000:]
    res = [np.load(os.path.join(str(tmp_path), "{}_r{}.npz".format(case, r))) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(res[0]["params"], res[r]["params"]), "replicas diverged ({}, rank {})".format(case, r)
        assert np.array_equal(res[0]["losses"], res[r]["losses"]), "ranks report different losses"
    return res


def _delta_err(got, ref, start):
    return np.linalg.norm((got - start) - (ref - start)) / (np.linalg.norm(ref - start) + 1e-30)


def _single_ppo(batch_size, feed_order, perms_of):
    """the single-process update of the same plugin pair: trajectories in `feed_order`, injected permutations"""
    import dp_plugin_worker as W
    from xingtian_amd.algorithm import alg_builder
    alg = alg_builder("PPO", W.ppo_model_info({"DP": "off", "BATCH_SIZE": batch_size}), {"instance_num": W.N_TRAJ, "agent_num": 1})
    start = alg.actor.net.params.cpu().numpy().copy()
    losses = []
    for u in range(W.UPDATES):
        trs = W.ppo_trajs(u)
        for k in feed_order:
            alg.prepare_data(trs[k])
        losses.append(float(alg.train(perms=perms_of(u))))
    torch.cuda.synchronize()
    return alg.actor.net.params.cpu().numpy(), start, losses


def _cumulative_perms(rng, n, epochs):
    inds = np.arange(n)
    out = np.empty((epochs, n), np.int32)
    for ep in range(epochs):
        rng.shuffle(inds)
        out[ep] = inds
    return out


def test_ppo_strict_replicated_feed_equals_the_single_gpu_update_and_rank0_publishes(tmp_path):
    """every rank is handed every trajectory, shared permutation seed, rank r takes its rows of every global minibatch of 32
    (16 + 16; the 96-row rollout has three full minibatches): the single-process update up to fp32 summation order; the
    reported loss is the global one; only rank 0 answers checkpoint_ready / if_save."""
    import dp_plugin_worker as W
    res = _run(tmp_path, "ppo-strict-replicated-torch")
    n = W.N_TRAJ * W.T_LEN
    rng = np.random.default_rng(W.PPO_MC["SEED"])          # the shared seed IS the configured one
    ref, start, losses = _single_ppo(32, range(W.N_TRAJ), lambda u: _cumulative_perms(rng, n, 2))
    assert _delta_err(res[0]["params"], ref, start) < 5e-3
    assert np.allclose(res[0]["losses"], losses, rtol=2e-3, atol=1e-5), (res[0]["losses"], losses)
    assert res[0]["answers"].all() and not res[1]["answers"].any()
    assert not res[1]["if_save"][0]                  # (rank 0 answers as configured, the others never save)


def test_ppo_strict_direct_exchange_in_the_update_graph_is_bitwise_the_gloo_hook_path(tmp_path):
    """DP_EXCHANGE direct: xt_allreduce_direct over hipIpc-mapped exchange blocks, captured into the hipGraph of
    xt_net_ppo_train and replayed -- a two-rank sum is the same in either order, so the parameters equal the gloo run's
    bit for bit."""
    a = _run(tmp_path, "ppo-strict-replicated-torch")
    b = _run(tmp_path, "ppo-strict-replicated-direct")
    assert np.array_equal(a[0]["params"], b[0]["params"])
    assert np.array_equal(a[0]["losses"], b[0]["losses"])


@pytest.mark.parametrize("feed", ["round_robin", "sharded"])
def test_ppo_strict_over_sharded_trajectories_equals_the_single_gpu_update_on_stratified_minibatches(tmp_path, feed):
    """trajectory k lives on rank k % 2 (kept at ingest from a replicated stream, or handed over that way); a global
    minibatch of 32 = 16 rows of each rank's LOCAL permutation.  Same update as one process that is fed rank 0's trajectories
    then rank 1's and whose minibatch k is the concatenation of the two local minibatches k."""
    import dp_plugin_worker as W
    res = _run(tmp_path, "ppo-strict-{}-torch".format(feed))
    nl = W.N_TRAJ // 2 * W.T_LEN                              # local rows (48): minibatches of 16 rows
    rngs = [np.random.default_rng([W.PPO_MC["SEED"], r]) for r in range(2)]
    order = [0, 2, 4, 1, 3, 5]

    def perms(u):
        loc = [_cumulative_perms(rngs[r], nl, 2) for r in range(2)]
        out = []
        for ep in range(2):
            row = []
            for s in range(0, nl, 16):
                row += list(loc[0][ep, s:s + 16]) + list(nl + loc[1][ep, s:s + 16])
            out.append(row)
        return np.asarray(out, np.int32)

    ref, start, losses = _single_ppo(32, order, perms)
    assert _delta_err(res[0]["params"], ref, start) < 5e-3
    assert np.allclose(res[0]["losses"], losses, rtol=2e-3, atol=1e-5)
    # each rank has its own explorers only in the `sharded` feed: there every rank publishes
    assert res[0]["answers"].all() and res[1]["answers"].all() == (feed == "sharded")


def test_ppo_weak_mode_trains_the_union_minibatch(tmp_path):
    """weak: every rank trains full 32-row minibatches of its own trajectories, gradients averaged: one process with
    BATCH_SIZE 64 on the union (flagged semantic change of the global batch)."""
    import dp_plugin_worker as W
    res = _run(tmp_path, "ppo-weak-sharded-torch")
    nl = W.N_TRAJ // 2 * W.T_LEN
    rngs = [np.random.default_rng([W.PPO_MC["SEED"], r]) for r in range(2)]

    def perms(u):
        loc = [_cumulative_perms(rngs[r], nl, 2) for r in range(2)]
        out = []
        for ep in range(2):
            row = []
            for s in range(0, nl, 32):
                row += list(loc[0][ep, s:s + 32]) + list(nl + loc[1][ep, s:s + 32])
            out.append(row)
        return np.asarray(out, np.int32)

    ref, start, losses = _single_ppo(64, [0, 2, 4, 1, 3, 5], perms)
    assert _delta_err(res[0]["params"], ref, start) < 5e-3
    assert np.allclose(res[0]["losses"], losses, rtol=2e-3, atol=1e-5)


def _single_impala(batch_size, order):
    import dp_plugin_worker as W
    from xingtian_amd.algorithm import alg_builder
    alg = alg_builder("IMPALAOpt", W.impala_model_info({"DP": "off"}), dict(W.IMPALA_ALG, BATCH_SIZE=batch_size))
    start = alg.actor.net.params.cpu().numpy().copy()
    losses = []
    for u in range(W.UPDATES):
        msgs = W.impala_msgs(u)
        for k in order:
            alg.prepare_data(msgs[k])
        losses.append(float(alg.train()))
    torch.cuda.synchronize()
    return alg.actor.net.params.cpu().numpy(), start, losses


@pytest.mark.parametrize("exchange", ["torch", "direct"])
def test_impala_strict_replicated_shards_whole_trajectories_inside_the_library(tmp_path, exchange):
    """IMPALAOpt, sum-form loss (impala_cnn_opt.py:299-351): every rank is handed every message; xt_net_impala_train takes
    this rank's whole-trajectory shard of every 40-frame chunk (4 trajectories -> 2 + 2), gradients SUMMED, no scaling =
    the single-process train; the logged loss is the global sum / chunks."""
    res = _run(tmp_path, "impala-strict-replicated-{}".format(exchange))
    ref, start, losses = _single_impala(40, range(4))
    assert _delta_err(res[0]["params"], ref, start) < 5e-3
    assert np.allclose(res[0]["losses"], losses, rtol=2e-3, atol=1e-4), (res[0]["losses"], losses)
    assert res[0]["answers"].all() and not res[1]["answers"].any()


def test_impala_strict_round_robin_messages_equal_the_single_gpu_chunks(tmp_path):
    """messages k % 2 == rank kept at ingest; a global chunk of 40 frames = 20 local frames (2 trajectories) per rank.
    One process fed [m0, m1, m2, m3] trains chunks {m0, m1}, {m2, m3}; the ranks train {m0}+{m1}, {m2}+{m3}: the same sums."""
    res = _run(tmp_path, "impala-strict-round_robin-torch")
    ref, start, losses = _single_impala(40, range(4))
    assert _delta_err(res[0]["params"], ref, start) < 5e-3
    assert np.allclose(res[0]["losses"], losses, rtol=2e-3, atol=1e-4)


def test_impala_weak_sums_the_ranks_full_chunks(tmp_path):
    """weak: every rank trains BATCH_SIZE = 40-frame chunks of its own messages: global chunk 80 frames (flagged)."""
    res = _run(tmp_path, "impala-weak-sharded-torch")
    ref, start, losses = _single_impala(80, [0, 2, 1, 3])
    assert _delta_err(res[0]["params"], ref, start) < 5e-3
    assert np.allclose(res[0]["losses"], losses, rtol=2e-3, atol=1e-4)


def test_direct_exchange_soak_25_updates_of_the_replayed_graph_stay_bitwise_on_the_gloo_path(tmp_path):
    """25 PPO updates (150 SGD steps = 150 all-reduces inside the replayed hipGraph, one launch each) through the plugin pair
    on two ranks: parameters and every reported loss bit for bit those of the host-synchronous gloo exchange -- a stale
    inbox / result read or a lost flag anywhere in the run would show."""
    a = _run(tmp_path, "ppo-strict-replicated-torch", updates=25)
    b = _run(tmp_path, "ppo-strict-replicated-direct", updates=25)
    assert len(a[0]["losses"]) == 25
    assert np.array_equal(a[0]["params"], b[0]["params"]) and np.array_equal(a[0]["losses"], b[0]["losses"])


def test_breakout_ppo_yaml_through_build_learner_algorithm_as_two_ranks(tmp_path):
    """INTEGRATION.md section 3b end to end: every rank runs ``build_learner_algorithm(<examples/breakout_ppo.yaml>)`` (the
    reference's own YAML, env_num 10 x 128 steps, BATCH_SIZE 320, 4 epochs = 16 SGD steps of 160 + 160 rows) with only the
    data-parallel keys added to ``model_config``; exchange = the direct all-reduce inside the replayed hipGraph.  Against the
    single-process learner built from the same YAML."""
    import dp_plugin_worker as W
    from xingtian_amd.config import build_learner_algorithm
    res = _run(tmp_path, "yaml-strict-replicated-direct")
    alg = build_learner_algorithm(W.yaml_config({"DP": "off"}))
    start = alg.actor.net.params.cpu().numpy().copy()
    losses = []
    for u in range(W.UPDATES):
        for tr in W.yaml_trajs(u, 10):
            alg.prepare_data(tr)
        losses.append(float(alg.train(episode_num=u)))
    torch.cuda.synchronize()
    ref = alg.actor.net.params.cpu().numpy()
    # 2 x 16 sign-like Adam steps on noise: the summation-order bar of the full-size update (test_gpu_learner: 0.15)
    assert _delta_err(res[0]["params"], ref, start) < 0.15, _delta_err(res[0]["params"], ref, start)
    assert np.allclose(res[0]["losses"], losses, rtol=5e-3, atol=1e-4), (res[0]["losses"], losses)
    assert res[0]["answers"].all() and not res[1]["answers"].any()
