"""GPU parity tests of the round-4 additions, all through the C ABI: ragged learner-side GAE, channel padding of
3-channel image observations (examples/ant_ppo.yaml), raw-trajectory ingest (GAE batched on the device), the packed
weight publish into a page-locked ring, the RCCL shim of the exchange hook, and the hook-routed data-parallel IMPALA
optimisers."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from oracle import nets, returns

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _d(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_gae_ragged_is_bit_exact_with_the_reference_loop():
    """xt_gae_f64_ragged on trajectories of different lengths laid out back to back (T = 1, 2, 37, 128, 200, 1024, 1025,
    1500: both the LDS form and the one-lane walk) against oracle.returns.gae, which is pinned bit-for-bit to the
    reference's PPO.data_proc (tests/golden/gae_*.npz)."""
    from xingtian_amd import lib as L
    lib = L.load()
    rng = np.random.default_rng(11)
    lens = [1, 2, 37, 128, 200, 1024, 1025, 1500, 128, 5]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    n = int(offs[-1])
    value_rows = np.empty(n, np.float32)
    boot = np.empty(len(lens), np.float32)
    reward = np.empty(n, np.float64)
    done = np.zeros(n, np.uint8)
    want_adv, want_tgt = np.empty(n), np.empty(n)
    for i, t in enumerate(lens):
        v = rng.standard_normal(t + 1).astype(np.float32)
        r = rng.choice([-1.0, 0.0, 1.0, 0.37], size=t)
        dn = rng.random(t) < (0.0 if i == 0 else 0.05)
        if i == 3:
            dn[:] = True
        if i == 4:
            dn[0] = dn[-1] = True
        a, ov, tg = returns.gae(v.reshape(-1, 1), r.copy(), dn)
        lo, hi = offs[i], offs[i + 1]
        value_rows[lo:hi], boot[i], reward[lo:hi], done[lo:hi] = v[:t], v[t], r, dn
        want_adv[lo:hi], want_tgt[lo:hi] = a[:, 0], tg[:, 0]
        assert np.array_equal(ov[:, 0], v[:t])
    adv = torch.full((n,), np.nan, dtype=torch.float64, device="cuda")
    tgt = torch.full((n,), np.nan, dtype=torch.float64, device="cuda")
    dv, db, dr, dd, do = _d(value_rows), _d(boot), _d(reward), _d(done), _d(offs)      # (kept alive across the launch)
    L.check(lib.xt_gae_f64_ragged(L.ptr(dv), L.ptr(db), L.ptr(dr), L.ptr(dd), L.ptr(do),
                                  L.ptr(adv), L.ptr(tgt), len(lens), 0.99, 0.95, L.stream_ptr()), "xt_gae_f64_ragged")
    torch.cuda.synchronize()
    assert np.array_equal(adv.cpu().numpy(), want_adv) and np.array_equal(tgt.cpu().numpy(), want_tgt)


def test_pad_channels_u8_and_f32():
    from xingtian_amd import lib as L
    lib = L.load()
    rng = np.random.default_rng(2)
    for dt, fill in ((np.uint8, 128), (np.uint8, 0), (np.float32, 0)):
        src = (rng.integers(0, 256, (7, 5, 3)) if dt == np.uint8 else rng.standard_normal((7, 5, 3))).astype(dt)
        dst = torch.empty((7, 5, 4), dtype=torch.uint8 if dt == np.uint8 else torch.float32, device="cuda")
        dsrc = _d(src)
        L.check(lib.xt_pad_channels(L.ptr(dsrc), L.ptr(dst), 35, 3, 4, src.itemsize, fill, L.stream_ptr()), "xt_pad_channels")
        got = dst.cpu().numpy()
        assert np.array_equal(got[..., :3], src) and np.all(got[..., 3] == (fill if dt == np.uint8 else 0))
    with pytest.raises(RuntimeError, match="elem_bytes"):
        L.check(lib.xt_pad_channels(L.ptr(dst), L.ptr(dst), 1, 3, 4, 2, 0, L.stream_ptr()), "xt_pad_channels")


@pytest.mark.parametrize("b", [10, 64])
def test_ppo_step_on_three_channel_frames_vs_oracle(b):
    """ant_ppo.yaml's shape (PpoCnn on [84, 84, 3] uint8, hidden 512, BATCH_SIZE 10): the padded HIP network against the
    float64 oracle of the UNPADDED network -- loss 1e-4, every gradient tensor 1e-5 (kernel gradient compared in the TF
    shape [8, 8, 3, 32]); the padded input-channel rows of the flat buffers stay exactly zero through the update."""
    from test_gpu_learner import PPO_CFG, assert_update_close, oracle_params_for, rel_err, synth_ppo_rollout
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    spec = netspec.ppo_cnn((84, 84, 3), 4, (512,), "relu", True)
    ospec = nets.ppo_cnn_spec((84, 84, 3), 4, (512,), "relu", True)
    net = HipActorCritic(spec, max_batch=b, seed=0)
    params = oracle_params_for(net, ospec, seed=7)
    rng = np.random.default_rng(0)
    n = b + 6
    obs, lab = synth_ppo_rollout(rng, n, (84, 84, 3), 4, True)
    idx = rng.permutation(n)[:b].astype(np.int32)
    cfg = dict(PPO_CFG, BATCH_SIZE=b)
    orc = nets.PpoLearnerOracle(ospec, params, cfg, np.float64)
    out = orc.step(obs[idx], lab[0][idx], lab[1][idx].astype(np.float32), lab[2][idx].astype(np.float32),
                   lab[3][idx].astype(np.float32), lab[4][idx].astype(np.float32), apply=True)
    dobs = net.to_device_obs(obs)
    assert tuple(dobs.shape) == (n, 84, 84, 4) and not dobs[..., 3].any()
    lo = net.ppo_step(net.make_ppo_cfg(cfg), dobs, _d(idx), _d(lab[0]), _d(lab[1].reshape(-1)), _d(lab[2].reshape(-1)),
                      _d(lab[3].reshape(-1)), _d(lab[4].reshape(-1)), apply=True)
    torch.cuda.synchronize()
    loss = lo.cpu().numpy()[0]
    assert abs(loss - out["loss"]) <= 1e-4 * max(1.0, abs(out["loss"]))
    g = net.grads_dict()
    assert g["shared_conv_layer_0/kernel"].shape == (8, 8, 3, 32)
    for k, ref in out["grads"].items():
        assert rel_err(g[k].reshape(ref.shape), ref) < 1e-5, k
    assert_update_close(net.get_weights(), orc.net.params, params, cfg["LR"], "cnn84_c3")
    off, size = spec.var_extent("shared_conv_layer_0/kernel")
    for buf in (net.params, net.grads, net.adam_m, net.adam_v):
        block = buf[off:off + size].cpu().numpy().reshape(8, 8, 4, 32)
        assert not block[:, :, 3].any()
    assert net.get_weights()["shared_conv_layer_0/kernel"].shape == (8, 8, 3, 32)


@pytest.mark.parametrize("rel", ["examples/ant_ppo.yaml", "examples/dog_ppo.yaml", "examples/beamrider_ppo.yaml",
                                 "examples/pong_ppo.yaml", "examples/qbert_ppo.yaml", "examples/spaceinvader_ppo.yaml",
                                 "examples/beamrider_impala.yaml", "examples/qbert_impala.yaml",
                                 "examples/spaceinvader_impala.yaml"])
def test_remaining_example_yamls_build_a_learner_and_train(rel):
    """The other PPO / IMPALA examples of the reference (tests/golden/learner_config.json now holds all 15) through
    xingtian_amd.config.build_learner_algorithm: one synthetic update through prepare_data (streaming ingest, incl. the
    channel padding of ant / dog), a finite loss, weights by TF name in the reference's shapes."""
    from xingtian_amd import config as cfg
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "learner_config.json")))[rel]
    conf = json.loads(json.dumps(g["config"]))
    conf["model_para"]["actor"].setdefault("model_config", {})
    conf["model_para"]["actor"]["model_config"]["SEED"] = 0
    alg = cfg.build_learner_algorithm(conf, g["env_info"])
    actor_info = conf["model_para"]["actor"]
    sd, ad = tuple(actor_info["state_dim"]), actor_info["action_dim"]
    rng = np.random.default_rng(3)
    if g["alg_para"]["alg_name"] == "PPO":
        t = 40
        for _ in range(alg.prepare_data_times):
            alg.prepare_data({"cur_state": rng.integers(0, 256, (t,) + sd).astype(np.uint8),
                              "action": rng.integers(0, ad, t).astype(np.int32), "logp": -np.ones((t, 1), np.float32),
                              "adv": rng.standard_normal((t, 1)), "old_value": rng.standard_normal((t, 1)).astype(np.float32),
                              "target_value": rng.standard_normal((t, 1))})
    else:
        tlen = actor_info["model_config"]["sample_batch_step"]
        envs = conf["env_para"]["env_info"].get("vector_env_size", 1)
        for _ in range(conf["alg_para"]["alg_config"]["prepare_times_per_train"]):
            n = tlen * envs
            alg.prepare_data({"cur_state": rng.integers(0, 256, (n,) + sd).astype(np.uint8),
                              "logit": rng.standard_normal((n, ad)).astype(np.float32),
                              "action": rng.integers(0, ad, n).astype(np.int32), "done": list(rng.random(n) < 0.05),
                              "reward": list(rng.choice([-1.0, 0.0, 1.0], n))})
    loss = alg.train(episode_num=0)
    assert isinstance(loss, (float, np.floating)) and np.isfinite(loss)
    w = alg.get_weights()
    first = next(iter(w.values()))
    assert first.shape[2] == sd[2] and all(isinstance(v, np.ndarray) for v in w.values())


def test_raw_trajectories_get_one_batched_gae_on_the_device_and_match_the_actor_side_path():
    """SURVEY 8(a1) on the learner: trajectories that arrive WITHOUT advantages (value [T+1], reward, done as the explorer
    holds them before data_proc) are streamed to HBM like any other, and ONE xt_gae_f64_ragged launch inside train()
    produces adv / target_v on the device -- no per-message launch, no read-back.  The update is bit-identical to the
    reference protocol (actor-side numpy GAE, shipped adv / old_value / target_value), lengths ragged."""
    from test_gpu_learner import synth_ppo_rollout
    from xingtian_amd.algorithm import alg_builder

    def mk():
        model_info = {"actor": {"model_name": "PpoCnn", "state_dim": [42, 42, 4], "action_dim": 3, "input_dtype": "uint8",
                                "model_config": {"BATCH_SIZE": 64, "NUM_SGD_ITER": 2, "hidden_sizes": [64], "SEED": 5,
                                                 "action_type": "Categorical", "USE_HIP_GRAPH": True}}}
        return alg_builder("PPO", model_info, {"instance_num": 5, "agent_num": 1})

    rng = np.random.default_rng(21)
    lens = [50, 37, 64, 1, 50]
    raw, cooked = [], []
    for t in lens:
        obs, lab = synth_ppo_rollout(rng, t, (42, 42, 4), 3)
        value = rng.standard_normal((t + 1, 1)).astype(np.float32)
        reward = rng.choice([-1.0, 0.0, 1.0], size=t)
        done = rng.random(t) < 0.05
        a, ov, tg = returns.gae(value, reward.copy(), done)
        raw.append({"cur_state": obs, "action": lab[0], "logp": lab[1], "value": value, "reward": list(reward), "done": list(done)})
        cooked.append({"cur_state": obs, "action": lab[0], "logp": lab[1], "adv": a, "old_value": ov, "target_value": tg})
    n = sum(lens)
    perms = np.stack([rng.permutation(n) for _ in range(2)]).astype(np.int32)
    res = []
    for msgs in (cooked, raw):
        alg = mk()
        for rep in range(2):            # the second update replays the captured graph on the other buffer set
            for m in msgs:
                alg.prepare_data(m)
            loss = alg.train(perms=perms)
        res.append((loss, alg.actor.net.params.cpu().numpy().copy()))
        if msgs is raw:
            d = alg.actor._ingest.last.dev
            want = np.concatenate([c["adv"][:, 0] for c in cooked])
            assert np.array_equal(d["adv"][:n].cpu().numpy(), want)
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    # a rollout must not mix the two kinds
    #  ... and the odd message is rejected AS IT ARRIVES, before any copy is enqueued for it: the rollout gathered
    # so far stays usable instead of poisoning every later train
    alg = mk()
    alg.prepare_data(raw[0])
    with pytest.raises(ValueError, match="must not mix"):
        alg.prepare_data(cooked[1])
    alg.prepare_data(raw[1])
    assert np.isfinite(alg.train())
    alg.prepare_data(cooked[0])                     # the next rollout may be of the other kind
    with pytest.raises(ValueError, match="must not mix"):
        alg.prepare_data(raw[1])
    alg.prepare_data(cooked[1])
    assert np.isfinite(alg.train())


def test_weights_publish_into_a_page_locked_ring_is_one_dma_and_readers_get_the_dict():
    """f2 fast path: ``Algorithm.publish_weights(ring)`` on a pinned WeightsRing copies the packed parameter block from
    HBM straight into the shared-memory slot (no host copy on the learner); a reader's ``fetch`` rebuilds the name-keyed
    dict -- equal to ``get_weights()``, also for a channel-padded first layer.  Unpinned rings take the per-variable copy."""
    from xingtian_amd import transport
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    for sd in ((84, 84, 4), (84, 84, 3)):
        net = HipActorCritic(netspec.ppo_cnn(sd, 4, (256,), "relu", True), max_batch=8, seed=3)
        for pinned in (True, False):
            want = net.get_weights()
            ring = transport.WeightsRing(slot_bytes=8 << 20, slots=3)
            try:
                if pinned:
                    assert ring.pin()
                k = net.publish_weights(ring, {"train_count": 7})
                reader = transport.WeightsRing(name=ring.name, slot_bytes=8 << 20, slots=3, create=False)
                seq, ctr, got = reader.fetch()
                assert seq == k == 1 and ctr["train_count"] == 7 and list(got) == list(want)
                for name in want:
                    assert got[name].shape == want[name].shape and np.array_equal(got[name], want[name]), name
                net.params.mul_(1.5)
                net.touch()
                assert net.publish_weights(ring) == 2
                _, _, got2 = reader.fetch()
                assert np.array_equal(got2["pi_latent/kernel"], net.get_weights()["pi_latent/kernel"])
                reader.close()
            finally:
                ring.close()
    w1 = net.get_weights()
    w2 = net.get_weights()
    keep = {k: a.copy() for k, a in w1.items()}
    assert all(not np.shares_memory(w1[k], w2[k]) for k in w1)        # public API: arrays nobody else holds
    for _ in range(2 * net.SNAP_SLOTS):                                 # ... that later snapshots never overwrite
        net.params.add_(1.0)
        net.touch()
        net.get_weights(copy=False)
    assert all(np.array_equal(w1[k], keep[k]) for k in w1)


def test_rccl_shim_exchange_on_one_rank_matches_the_stepwise_path_bitwise():
    """xt_net_set_rccl: the library calls ncclAllReduce itself through the function pointer (no Python trampoline).  A
    1-rank communicator (all a 1-GPU box can form) must reproduce the step-wise data-parallel path bit for bit -- eager
    enqueue, captured into the update's hipGraph, and with the two-bucket overlap -- and the status word counts the calls."""
    from test_gpu_learner import synth_ppo_rollout
    from xingtian_amd import parallel
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    spec = netspec.ppo_cnn((42, 42, 4), 4, (64,), "relu", True)
    cfg = dict(LR=2.5e-4, LOSS_CLIPPING=0.1, ENTROPY_LOSS=0.003, VF_CLIP=5.0, CRITIC_LOSS_COEF=1.0, MAX_GRAD_NORM=5.0,
               BATCH_SIZE=32, NUM_SGD_ITER=2)
    rng = np.random.default_rng(4)
    n = 80
    obs, lab = synth_ppo_rollout(rng, n, (42, 42, 4), 4)
    perms = np.stack([rng.permutation(n) for _ in range(2)]).astype(np.int32)
    args = lambda net: (net.to_device_obs(obs), _d(perms), _d(lab[0]), _d(lab[1].reshape(-1)), _d(lab[2].reshape(-1)),
                        _d(lab[3].reshape(-1)), _d(lab[4].reshape(-1)))
    ref = HipActorCritic(spec, max_batch=32, seed=5)
    parallel.dp_ppo_update(ref, cfg, *args(ref), 0, 1, mode="weak")
    torch.cuda.synchronize()
    want = ref.params.cpu().numpy()
    try:
        comm = parallel.RcclComm(0, 1)
    except Exception as exc:      # noqa: BLE001
        pytest.skip("no RCCL communicator on this box: %r" % (exc,))
    from xingtian_amd import lib as L
    warm = torch.zeros(256, dtype=torch.float32, device="cuda")
    comm.all_reduce_(warm, L.stream_ptr())
    torch.cuda.synchronize()
    for graph, overlap in ((False, False), (True, False), (False, True)):
        net = HipActorCritic(spec, max_batch=32, seed=5)
        comm.attach(net, overlap=overlap)
        net.ppo_train(net.make_ppo_cfg(cfg, grad_scale=1.0), *args(net), use_graph=graph)
        torch.cuda.synchronize()
        calls, err = comm.status(net)
        comm.detach(net)
        assert err == 0 and calls == 6 * (2 if overlap else 1), (graph, overlap, calls, err)
        assert np.array_equal(net.params.cpu().numpy(), want), (graph, overlap)
    comm.destroy()


def test_async_loss_option_lags_the_reported_loss_by_one_train_and_nothing_else():
    """model_config ASYNC_LOSS (IMPALAOpt): train() returns the previous train's loss without waiting for the update it
    enqueued (the first call waits for its own), so the host stages the next message while the GPU works.  Same weights
    after every train, same losses shifted by one."""
    from xingtian_amd.algorithm import alg_builder

    def mk(async_loss):
        model_info = {"actor": {"model_name": "ImpalaCnnOpt", "state_dim": [42, 42, 4], "input_dtype": "uint8",
                                "state_mean": 128.0, "state_std": 128.0, "action_dim": 6,
                                "model_config": {"LR": 1e-3, "sample_batch_step": 10, "grad_norm_clip": 40.0, "SEED": 0,
                                                 "ASYNC_LOSS": async_loss, "lr_schedule": [[0, 1e-3], [20000, 1e-6]]}}}
        return alg_builder("IMPALAOpt", model_info, {"instance_num": 4, "agent_num": 1, "prepare_times_per_train": 1,
                                                    "train_per_checkpoint": 3, "BATCH_SIZE": 50})

    rng = np.random.default_rng(9)
    msgs = []
    for _ in range(5):
        n = 50
        msgs.append({"cur_state": rng.integers(0, 256, (n, 42, 42, 4)).astype(np.uint8),
                     "logit": rng.standard_normal((n, 6)).astype(np.float32), "action": rng.integers(0, 6, n).astype(np.int32),
                     "done": list(rng.random(n) < 0.05), "reward": list(rng.choice([-1.0, 0.0, 1.0], n))})
    out = {}
    for mode in (False, True):
        alg = mk(mode)
        losses = []
        for m in msgs:
            alg.prepare_data(m)
            losses.append(float(alg.train()))
        out[mode] = (losses, alg.get_weights())
    sync, lag = out[False][0], out[True][0]
    assert lag[0] == sync[0] and lag[1:] == sync[:-1], (sync, lag)
    for k in out[False][1]:
        assert np.array_equal(out[False][1][k], out[True][1][k]), k


def test_publish_weights_with_a_lag_of_one_hands_out_the_previous_update_without_waiting():
    """publish_weights(ring, lag=1): after update k the readers get the weights of update k-1 (whose D2H landed long ago);
    the copy of update k stays begun in the ring's next slot.  lag=0 afterwards catches up to the current weights."""
    from xingtian_amd import transport
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    net = HipActorCritic(netspec.ppo_mlp((4,), 2, (16, 16), "tanh", True), max_batch=8, seed=0)
    ring = transport.WeightsRing(slot_bytes=1 << 20, slots=3)
    assert ring.pin()
    reader = transport.WeightsRing(name=ring.name, slot_bytes=1 << 20, slots=3, create=False)
    try:
        net.attach_weights_ring(ring)
        hist = []
        for k in range(5):
            net.params.add_(1.0)            # "update k"
            net.touch()
            net.snapshot_weights_async()    # what every train enqueues
            hist.append(net.params.cpu().numpy().copy())
            seq = net.publish_weights(ring, lag=1)
            got = reader.fetch(newer_than=0)
            flat = np.concatenate([got[2][n].reshape(-1) for n in net.spec.names])
            want = hist[max(k - 1, 0)]
            ref = np.concatenate([want[off:off + int(np.prod(shape))] for off, shape in net.spec.names.values()])
            assert np.array_equal(flat, ref), (k, seq)
        assert net.publish_weights(ring, lag=0) == reader.latest()
        got = reader.fetch(newer_than=0)
        flat = np.concatenate([got[2][n].reshape(-1) for n in net.spec.names])
        ref = np.concatenate([hist[-1][off:off + int(np.prod(shape))] for off, shape in net.spec.names.values()])
        assert np.array_equal(flat, ref)
    finally:
        reader.close()
        ring.close()


def test_gae_ragged_degenerate_inputs():
    """no trajectories at all, and zero-length trajectories between real ones: nothing is written for them, the others
    are exact (the reference never ships an empty trajectory; the ingest must not trip over one)."""
    from xingtian_amd import lib as L
    lib = L.load()
    z = torch.zeros(4, dtype=torch.float32, device="cuda")
    z64 = torch.zeros(4, dtype=torch.float64, device="cuda")
    zu = torch.zeros(4, dtype=torch.uint8, device="cuda")
    zi = torch.zeros(4, dtype=torch.int32, device="cuda")
    L.check(lib.xt_gae_f64_ragged(L.ptr(z), L.ptr(z), L.ptr(z64), L.ptr(zu), L.ptr(zi), L.ptr(z64), L.ptr(z64), 0, 0.99, 0.95,
                                  L.stream_ptr()), "xt_gae_f64_ragged")
    rng = np.random.default_rng(3)
    lens = [0, 5, 0, 0, 3, 0]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    n = int(offs[-1])
    vr, boot = rng.standard_normal(n).astype(np.float32), rng.standard_normal(len(lens)).astype(np.float32)
    rew, dn = rng.standard_normal(n), (rng.random(n) < 0.3).astype(np.uint8)
    want_a, want_t = np.empty(n), np.empty(n)
    for i, t in enumerate(lens):
        if t:
            lo, hi = offs[i], offs[i + 1]
            a, _, tg = returns.gae(np.concatenate([vr[lo:hi], boot[i:i + 1]]).reshape(-1, 1), rew[lo:hi].copy(), dn[lo:hi].astype(bool))
            want_a[lo:hi], want_t[lo:hi] = a[:, 0], tg[:, 0]
    adv = torch.full((n,), np.nan, dtype=torch.float64, device="cuda")
    tgt = torch.full((n,), np.nan, dtype=torch.float64, device="cuda")
    keep = [_d(vr), _d(boot), _d(rew), _d(dn), _d(offs)]
    L.check(lib.xt_gae_f64_ragged(*[L.ptr(k) for k in keep], L.ptr(adv), L.ptr(tgt), len(lens), 0.99, 0.95, L.stream_ptr()),
            "xt_gae_f64_ragged")
    torch.cuda.synchronize()
    assert np.array_equal(adv.cpu().numpy(), want_a) and np.array_equal(tgt.cpu().numpy(), want_t)


def test_cartpole_impala_reward_curve_through_plugins():
    """examples/cartpole_impala.yaml closed loop (tools/cartpole_impala_e2e.py): IMPALA + ImpalaMlp, explorers = the numpy
    replica sampling from the action probabilities, host v-trace from probabilities, Keras-form fit on the GPU, weights
    every second train.  20 Adam steps per 2000 env steps: slow, but the policy must improve (measured: 18.8 -> ~90 at
    round 150 with this seed; the run is deterministic)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import cartpole_impala_e2e
    state = np.random.get_state()
    try:
        curve = cartpole_impala_e2e.run(rounds=150, seed=0, verbose=False)
    finally:
        np.random.set_state(state)
    first, last = float(np.nanmean(curve[:5])), float(np.nanmean(curve[-20:]))
    assert first < 40.0, curve[:5]
    assert last > 45.0 and last > 2.0 * first, (first, last)


def test_pixel_control_reward_curve_through_the_headline_network():
    """tools/pixel_catch_e2e.py: breakout_ppo.yaml's model section (PpoCnn, 84x84x4 uint8 stacks, BATCH_SIZE 320, 4 epochs)
    in a closed loop with a synthetic Atari-shaped game: 32 raw 128-step uint8 trajectories per update through
    prepare_data, GAE on the GPU, the replayed 52-step graph, weights by name to a second PpoCnn that plays.  Random play
    scores about -0.7; measured with this seed: +0.25 after 12 updates, +0.9 after 24 (deterministic)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import pixel_catch_e2e
    curve = pixel_catch_e2e.run(updates=30, env_num=32, seed=0, verbose=False)
    first, last = float(np.mean(curve[:3])), float(np.mean(curve[-3:]))
    assert first < -0.4, curve[:3]
    assert last > 0.5, curve


def test_pixel_control_reward_curve_impala_vtrace_on_gpu():
    """tools/pixel_catch_impala_e2e.py: breakout_impala.yaml's sections (IMPALAOpt + ImpalaCnnOpt, 84x84x4 uint8, T = 128,
    one train per message) in a closed loop with the synthetic catch game; the 32 messages of a round are up to 31 trains
    stale (v-trace's job).  Random play about -0.7; measured with this seed: +0.38 at round 16, +0.95 at round 24."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import pixel_catch_impala_e2e
    curve = pixel_catch_impala_e2e.run(rounds=30, seed=0, verbose=False)
    first, last = float(np.mean(curve[:3])), float(np.mean(curve[-3:]))
    assert first < -0.4, curve[:3]
    assert last > 0.4, curve
