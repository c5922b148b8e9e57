"""``parallel.LearnerDP`` (data parallelism behind the plugin pair) without a GPU: two gloo ranks exercise what the model
constructors and the algorithms call on it -- configuration from the launcher environment, the minibatch split arithmetic of
every mode x feed, the round-robin ingest filter, the shared permutation seed and the arithmetic of the data-parallel tail
(rows + loss shares travelling with the gradient instead of two host collectives per train).  The GPU half is tests/test_gpu_dp_plugin.py."""
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
        # the data-parallel TAIL (include/xt_mi355x.h `xt_net_set_dp`) restated on the host: every rank fills slot [rank] with
        # its rows and slot [16 + rank] with its loss share, zeros elsewhere; after ONE SUM all-reduce every rank holds every
        # rank's values exactly (one non-zero summand per slot), derives the same global loss in rank order and sees a row
        # mismatch -- what replaced the two host collectives per train (row-count check, global loss) of ABI 10
        import torch
        from xingtian_amd.lib import DP_TAIL_FLOATS
        for rows_of in (lambda r: 4096.0, lambda r: 4096.0 + r):
            tail = torch.zeros(DP_TAIL_FLOATS, dtype=torch.float32)
            tail[rank], tail[16 + rank] = float(rows_of(rank)), float(np.float32(1.5 + 0.1 * rank))
            dist.all_reduce(tail)
            t = tail.numpy()
            assert [t[r] for r in range(world)] == [np.float32(rows_of(r)) for r in range(world)]
            loss = np.float32(0.0)
            for r in range(world):
                loss = np.float32(loss + t[16 + r])
            want = np.float32(0.0)
            for r in range(world):
                want = np.float32(want + np.float32(1.5 + 0.1 * r))
            assert loss == want and not t[world:16].any() and not t[16 + world:].any()
            assert (len({float(t[r]) for r in range(world)}) > 1) == (rows_of(1) != rows_of(0))
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


def test_gradient_entries_and_the_tail_tile_the_exchanged_buffer():
    """The gradient-reduction launch of a fused data-parallel step (csrc/xt_optim.hip, DpFinish.scatter) writes every entry's
    float4s (entries start 16-byte aligned; a short last vector is zero padded) and the 8 tail vectors straight into the
    owners' inboxes; nothing else ever writes the inboxes of a step.  For the real layouts: the entries' vector ranges and
    the tail are disjoint, lie inside the exchanged buffer and COVER it (an uncovered vector would carry a stale value into
    the sum), and every rank's slice is non-empty for every world size the comm accepts."""
    from xingtian_amd.lib import DP_TAIL_FLOATS
    from xingtian_amd.model import netspec
    specs = [netspec.ppo_cnn((84, 84, 4), 4, (256,), "relu", True), netspec.impala_cnn_opt((42, 42, 4), 6, 128.0, 128.0, "uint8"),
             netspec.ppo_mlp((4,), 2, (64, 64), "tanh", False), netspec.ppo_cnn((84, 84, 3), 4, (256,), "relu", True)]
    for spec in specs:
        nvec = (spec.n_flat + 3) // 4 + DP_TAIL_FLOATS // 4
        # kernel + bias of a layer are ONE entry of the gradient table: merge neighbours that touch
        merged = []
        for off, cnt in sorted(spec.var_extent(name) for name in spec.names):
            if merged and merged[-1][0] + merged[-1][1] == off:
                merged[-1][1] += cnt
            else:
                merged.append([off, cnt])
        covered = np.zeros(nvec, np.int32)
        for off, cnt in merged:
            assert off % 4 == 0 and off + cnt <= spec.n_flat
            covered[off // 4:off // 4 + (cnt + 3) // 4] += 1
        covered[nvec - DP_TAIL_FLOATS // 4:] += 1
        assert (covered == 1).all(), "vectors written {} times: {}".format(set(covered.tolist()), spec.state_dim)
        for world in range(1, 17):
            base, rem = divmod(nvec, world)
            assert base >= 1
