"""``parallel.LearnerDP`` (data parallelism behind the plugin pair) without a GPU: two gloo ranks exercise what the model
constructors and the algorithms call on it -- configuration from the launcher environment, the minibatch split arithmetic of
every mode x feed, the round-robin ingest filter, the shared permutation seed, the row-count guard (must fail on EVERY rank,
not dead-lock) and the global loss.  The GPU half is tests/test_gpu_dp_plugin.py."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _FakeNet(object):
    """records what LearnerDP asks of HipActorCritic.make_ppo_cfg"""

    def make_ppo_cfg(self, cfg, grad_scale=1.0, global_batch=0, shard_rank=0, shard_world=0):
        return dict(batch=int(cfg["BATCH_SIZE"]), grad_scale=grad_scale, global_batch=global_batch, shard_rank=shard_rank,
                    shard_world=shard_world)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from xingtian_amd.parallel import LearnerDP
    try:
        # WORLD_SIZE > 1 and no DP key: strict over a replicated feed, rank 0 publishes, group created on demand
        dp = LearnerDP.from_config({"DP_BACKEND": "gloo", "DP_EXCHANGE": "torch"})
        assert dist.is_initialized() and (dp.rank, dp.world, dp.mode, dp.feed) == (rank, world, "strict", "replicated")
        assert dp.is_publisher == (rank == 0) and not dp.graph_capable
        cfg = dict(BATCH_SIZE=320)
        c, local = dp.ppo_cfg(_FakeNet(), cfg)
        assert c == dict(batch=320, grad_scale=1.0, global_batch=0, shard_rank=rank, shard_world=world) and local == 320
        assert [dp.takes() for _ in range(4)] == [True] * 4
        # strict over sharded trajectories: BATCH_SIZE / N local rows, means over the global 320
        rr = LearnerDP.from_config({"DP": "strict", "DP_FEED": "round_robin", "DP_EXCHANGE": "direct"})
        c, local = rr.ppo_cfg(_FakeNet(), cfg)
        assert c == dict(batch=160, grad_scale=1.0, global_batch=320, shard_rank=0, shard_world=0) and local == 160
        assert [rr.takes() for _ in range(5)] == [k % world == rank for k in range(5)]
        rr.new_rollout()
        assert rr.takes() == (rank == 0) and rr.is_publisher == (rank == 0) and rr.graph_capable
        # RCCL inside a replayed hipGraph has only met one rank here: eager enqueue unless DP_GRAPH opts in
        assert not LearnerDP.from_config({"DP_EXCHANGE": "rccl"}).graph_capable
        assert LearnerDP.from_config({"DP_EXCHANGE": "rccl", "DP_GRAPH": True}).graph_capable
        assert not LearnerDP.from_config({"DP_EXCHANGE": "direct", "DP_GRAPH": False}).graph_capable
        with pytest.raises(ValueError, match="divisible"):
            rr.ppo_cfg(_FakeNet(), dict(BATCH_SIZE=321))
        # weak: full local minibatches, gradients averaged; every rank serves its own explorers
        wk = LearnerDP.from_config({"DP": "weak"})
        c, local = wk.ppo_cfg(_FakeNet(), cfg)
        assert c["batch"] == 320 and c["grad_scale"] == 1.0 / world and c["global_batch"] == 0 and wk.feed == "sharded"
        assert wk.is_publisher and LearnerDP.from_config({"DP": "weak", "DP_PUBLISH": "rank0"}).is_publisher == (rank == 0)
        assert LearnerDP.from_config({"DP": "off"}) is None
        # a model that is not the learner's never becomes a replica implicitly (attaching is a collective), only on request
        assert LearnerDP.from_config({}, is_learner=False) is None
        assert LearnerDP.from_config({"DP": "weak"}, is_learner=False).mode == "weak"
        for bad in ({"DP": "weak", "DP_FEED": "replicated"}, {"DP": "sideways"}, {"DP_FEED": "x"}, {"DP_EXCHANGE": "mpi"},
                    {"DP_PUBLISH": "nobody"}):
            with pytest.raises(ValueError):
                LearnerDP.from_config(bad)
        # one seed for the generators that must agree (rank 0's when none is configured), the configured one otherwise
        assert dp.shared_seed(7) == 7
        s = dp.shared_seed(None)
        seeds = [None] * world
        dist.all_gather_object(seeds, s)
        assert len(set(seeds)) == 1
        # the row-count guard raises on EVERY rank when the ranks disagree, and passes when they agree
        dp.check_equal(4096, "PPO.train")
        with pytest.raises(ValueError, match="different amounts of data"):
            dp.check_equal(4096 + rank, "PPO.train")
        # the loss the learner logs: strict = SUM of the ranks' shares / minibatches
        assert abs(dp.global_loss(1.5 + rank, 4) - sum(1.5 + r for r in range(world)) / 4) < 1e-12
        out[rank] = 1
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_learner_dp_configuration_split_arithmetic_and_guards_gloo_world2():
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: 1, 1: 1}


def test_learner_dp_is_off_in_a_single_process():
    from xingtian_amd.parallel import LearnerDP
    env = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        assert LearnerDP.from_config({}) is None
        assert LearnerDP.from_config({"DP": "strict"}) is None        # asked for, but nobody to exchange with
        assert LearnerDP.from_config(None) is None
    finally:
        for k, v in env.items():
            if v is not None:
                os.environ[k] = v


def test_direct_allreduce_slice_arithmetic_is_consistent_for_every_size_and_world():
    """The index arithmetic of csrc/xt_xgmi.hip restated in Python and checked exhaustively: ``slice_of`` (balanced contiguous
    split of the float4 vectors, the first ``rem`` slices one longer) and the owner-of-vector formula of the fused kernel's
    scatter phase (``v < cut ? v / (base + 1) : rem + (v - cut) / base``) must agree for every vector of every (count, world),
    the slices must tile [0, nvec) in rank order, an inbox slot (slice_cap) must hold the longest slice, and the per-slice
    vector tickets must add up to the slice lengths (the flag of a slice is raised by whoever completes its count)."""
    def slice_of(nvec, r, world):
        base, rem = divmod(nvec, world)
        b = r * base + min(r, rem)
        return b, b + base + (1 if r < rem else 0)

    def owner(v, nvec, world):
        base, rem = divmod(nvec, world)
        cut = rem * (base + 1)
        return v // (base + 1) if v < cut else rem + ((v - cut) // base if base else 0)

    for world in range(1, 17):
        for count in list(range(1, 200)) + [847496, 1005109, 250007, 4099]:
            nvec = (count + 3) // 4
            slice_cap = ((nvec + world - 1) // world) * 4                      # floats per inbox slot (xt_direct_create)
            edges = [slice_of(nvec, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == nvec
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            assert max(e - b for b, e in edges) * 4 <= slice_cap
            if nvec <= 4096:
                tickets = [0] * world
                for v in range(nvec):
                    q = owner(v, nvec, world)
                    assert edges[q][0] <= v < edges[q][1], (count, world, v, q)
                    tickets[q] += 1
                assert tickets == [e - b for b, e in edges]
            else:                                                             # the big sizes: the slice boundaries only
                for b, e in edges:
                    for v in {b, e - 1} if e > b else set():
                        assert owner(v, nvec, world) == edges.index((b, e))
