"""The data-parallel SGD step without host collectives and without gradient copies (C ABI 11: ``xt_net_set_dp`` +
``xt_net_set_direct``) on one process: the fused direct exchange as a one-rank group against the plain
single-GPU update, the error path (a peer that never shows up: the optimiser SKIPS the update and the error bits travel with
the loss), the collective reset, and the tail through the generic hook.  Multi-rank runs: tests/test_gpu_dp_ranks.py,
tests/test_gpu_dp_plugin.py.  Reference analogue of the message: xt/framework/trainer.py:139-144 (dead code there)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu

CFG = dict(LR=2.5e-4, LOSS_CLIPPING=0.1, ENTROPY_LOSS=0.003, VF_CLIP=5.0, CRITIC_LOSS_COEF=1.0, MAX_GRAD_NORM=5.0,
           BATCH_SIZE=32, NUM_SGD_ITER=2)


def _net_and_rollout(seed=3, n=80):
    from test_gpu_learner import synth_ppo_rollout
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    spec = netspec.ppo_cnn((42, 42, 4), 4, (64,), "relu", True)
    net = HipActorCritic(spec, max_batch=CFG["BATCH_SIZE"], seed=seed)
    rng = np.random.default_rng(17)
    obs, lab = synth_ppo_rollout(rng, n, (42, 42, 4), 4)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    perm = d(np.stack([rng.permutation(n) for _ in range(CFG["NUM_SGD_ITER"])]).astype(np.int32))
    args = (d(obs), perm, d(lab[0]), d(lab[1].reshape(-1)), d(lab[2].reshape(-1)), d(lab[3].reshape(-1)), d(lab[4].reshape(-1)))
    return net, args


@pytest.mark.parametrize("use_graph", [False, True])
def test_fused_direct_step_as_a_one_rank_group_reproduces_the_single_gpu_update(use_graph):
    """every kernel of the fused chain runs against the rank's own exchange block (gradient reduction -> inbox, reduce launch
    -> result + squared-norm partials, Adam <- result): the same gradient bit for bit (a one-rank sum adds nothing), the same
    loss, parameters to the rounding of a differently blocked norm sum; replaying the captured graph is bitwise the eager run"""
    from xingtian_amd.parallel import DirectComm
    ref, args = _net_and_rollout()
    acc = ref.ppo_train(ref.make_ppo_cfg(CFG), *args, use_graph=False)
    want_loss = acc.cpu().numpy()[:2].copy()
    want = ref.params.cpu().numpy()
    net, args = _net_and_rollout()
    start = net.params.cpu().numpy().copy()
    comm = DirectComm(0, 1, int(net.grads_xchg.numel()))
    net.set_dp(0, 1, 1.0)
    comm.attach_fused(net)
    for rep in range(2 if use_graph else 1):       # (graph: capture, then replay from the same start)
        net.params.copy_(torch.from_numpy(start)); net.reset_optimizer()
        acc = net.ppo_train(net.make_ppo_cfg(CFG), *args, use_graph=use_graph)
        a = net.read_loss(acc)
        got = net.params.cpu().numpy()
        assert np.array_equal(a[:2], want_loss), (a, want_loss)
        assert a[2] == 0.0
        err = np.linalg.norm(got - want) / np.linalg.norm(want - start)
        assert err < 1e-4, err
        if rep == 0:
            first = got.copy()
        else:
            assert np.array_equal(first, got)
    st = comm.status()
    assert st["error_bits"] == 0 and st["seq"] == (2 if use_graph else 1) * 6          # 3 minibatches x 2 epochs per update
    # the exchanged gradient of the last step == what the plain path leaves in net.grads for the same step
    g = comm.read_result(net.spec.n_flat)
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    comm.detach(net)
    net.set_dp(0, 0)
    comm.destroy()


def test_a_missing_peer_sets_the_error_bits_and_the_optimiser_skips_the_update():
    """two-rank group, rank 1 never steps: rank 0's reduce launch waits 50 ms for rank 1's slices, sets the sticky error word,
    every later wait returns at once, Adam SKIPS every update (parameters and slots bit for bit those before the train) and the
    bits arrive with the loss: read_loss raises.  After the (collective) reset the same comm works again."""
    from xingtian_amd.parallel import DirectComm
    net, args = _net_and_rollout()
    ranks = DirectComm.local_group(2, int(net.grads_xchg.numel()), timeout_ms=50)
    net.set_dp(0, 2, 1.0)
    ranks[0].attach_fused(net)
    start, m0 = net.params.cpu().numpy().copy(), net.adam_m.cpu().numpy().copy()
    acc = net.ppo_train(net.make_ppo_cfg(CFG, grad_scale=1.0, global_batch=0, shard_rank=0, shard_world=2), *args, use_graph=False)
    with pytest.raises(RuntimeError, match="never arrived"):
        net.read_loss(acc)
    assert np.array_equal(net.params.cpu().numpy(), start) and np.array_equal(net.adam_m.cpu().numpy(), m0)
    assert ranks[0].status()["error_bits"] & 1
    for c in ranks:
        c._L.check(c.lib.xt_direct_reset(c.comm), "xt_direct_reset")
    assert ranks[0].status() == dict(calls=ranks[0].status()["calls"], seq=0, error_bits=0)
    ranks[0].detach(net)
    net.set_dp(0, 0)
    for c in ranks:
        c.destroy()


def test_direct_info_counts_the_ranks_that_share_the_device():
    from xingtian_amd.parallel import DirectComm
    for world in (1, 2, 8):
        ranks = DirectComm.local_group(world, 4096)
        for c in ranks:
            info = c.info()
            assert info["ranks_on_device"] == world and info["block_cap"] >= 128 // world and info["block_cap"] * world <= 4096
        for c in ranks:
            c.destroy()
