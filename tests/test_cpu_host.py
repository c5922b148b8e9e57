"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, the plugin registry and
config surface behave like the reference's, network specs agree with the oracle's, and the data-parallel
plumbing is correct under a 2-process gloo group."""
import os
import re
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_loads_and_exports_every_header_symbol():
    from xingtian_amd import lib
    handle = lib.load()
    header = open(os.path.join(ROOT, "include", "xt_mi355x.h")).read()
    declared = set(re.findall(r"\b(xt_[a-z0-9_]+)\s*\(", header))
    declared -= {"xt_net_desc", "xt_layer_desc"}
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(handle, name), "missing C-ABI symbol " + name
        assert name in lib.SIGNATURES, "no ctypes prototype for " + name
    assert handle.xt_abi_version() == lib.ABI_VERSION == 12
    assert handle.xt_build_arch() == b"gfx950"
    # the binary carries the digest of the sources it was compiled from: a stale prebuilt .so next to newer kernel
    # sources is caught here (and profile artefacts are tagged with THIS digest, not with the tree's)
    assert lib.built_sources_sha() == lib.kernel_sources_sha(), "libxt_mi355x.so is stale: rebuild (make -C xingtian_amd/csrc)"


def test_tuning_struct_replaces_the_environment_switches():
    """Kernel-selection knobs live in ONE struct on the C ABI (xt_tuning_get / xt_tuning_set), defaults = the
    measured-best forms; the shipped library reads no environment variable."""
    from xingtian_amd import lib
    t = lib.get_tuning()
    assert t["bf16x6"] == 1 and t["conv1_bf16x3"] == 1 and t["conv1_waves"] == 8 and t["bwd_fit_slots"] == 768
    assert t["fwd_split_target"] == 256 and t["wgrad_split_target"] == 512 and t["finalize_ticket"] == 0 and t["fwd_tiled_valid"] == 1 \
        and t["wgrad_rows"] == 4 and t["fwd_prefetch_all"] == 0 and t["fwd_xcd_chunk"] == 1 and t["bwd_deep_prefetch"] == 1 and t["fwd_four_groups"] == 1 and t["reduce_deep_lanes"] == 128 and t["tail_overlap"] == 0 and t["tail_fused"] == 0
    old = lib.set_tuning(bf16x6=0, direct_waves=1024)
    try:
        assert old == {"bf16x6": 1, "direct_waves": 1536}
        t2 = lib.get_tuning()
        assert t2["bf16x6"] == 0 and t2["direct_waves"] == 1024 and t2["conv1_flat"] == t["conv1_flat"]
        with pytest.raises(RuntimeError, match="conv1_waves"):
            lib.set_tuning(conv1_waves=5)
        with pytest.raises(KeyError):
            lib.set_tuning(no_such_knob=1)
    finally:
        lib.set_tuning(**old)
    assert lib.get_tuning() == t
    for f in os.listdir(os.path.join(ROOT, "xingtian_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(ROOT, "xingtian_amd", "csrc", f)).read(), f


def test_product_path_fails_loudly_without_gpu():
    """The LEARNER (model_info type 'learner', xt/framework/learner.py:544) has no CPU fallback; a model built in an
    explorer / evaluator process (no GPU visible, no 'type') is the inference-only numpy replica: it predicts and
    takes weights by name but refuses to train."""
    from xingtian_amd import lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        lib.require_gpu()
    from xingtian_amd.model import model_builder
    info = {"model_name": "PpoMlp", "state_dim": [4], "action_dim": 2, "model_config": {"action_type": "Categorical"}}
    with pytest.raises(RuntimeError, match="no HIP device"):
        model_builder(dict(info, type="learner"))
    with pytest.raises(RuntimeError, match="no HIP device"):
        model_builder(dict(info, model_config={"action_type": "Categorical", "DEVICE": "gpu"}))
    from xingtian_amd.algorithm import alg_builder
    with pytest.raises(RuntimeError, match="no HIP device"):
        alg_builder("PPO", {"actor": dict(info, type="learner")}, {"instance_num": 1, "agent_num": 1})
    actor = model_builder(info)
    assert actor.net.inference_only
    state = np.zeros((3, 4), np.float32)
    action, logp, value = actor.predict(state)
    assert action.shape == (3,) and action.dtype == np.int32 and logp.shape == (3, 1) and value.shape == (3, 1)
    with pytest.raises(RuntimeError, match="inference-only"):
        actor.train([state], [action, logp, value, value, value])


@pytest.mark.parametrize("which", ["ppo_cnn84", "ppo_cnn42_unshared", "ppo_mlp", "impala84", "impala42", "pendulum"])
def test_cpu_replica_forward_matches_the_oracle_and_keeps_the_weight_contract(which):
    """the numpy replica explorers run (xingtian_amd/model/cpu_net.py -- product code, no oracle
    import) against the float64 oracle forward: NHWC im2col, VALID and TensorFlow's asymmetric SAME padding, Flatten
    in (H, W, C) order, the 11x11 conv as a dense layer, zero-padded odd feature counts; weights go in and out by TF
    variable name with TFVariables' error behaviour."""
    from xingtian_amd.model import netspec
    from xingtian_amd.model.cpu_net import CpuActorCritic
    rng = np.random.default_rng(5)
    if which == "ppo_cnn84":
        spec, ospec = netspec.ppo_cnn((84, 84, 4), 4, (256,), "relu", True), nets.ppo_cnn_spec((84, 84, 4), 4, (256,), "relu", True)
        obs = rng.integers(0, 256, (5, 84, 84, 4)).astype(np.uint8)
    elif which == "ppo_cnn42_unshared":
        spec, ospec = netspec.ppo_cnn((42, 42, 4), 6, (64,), "tanh", False), nets.ppo_cnn_spec((42, 42, 4), 6, (64,), "tanh", False)
        obs = rng.integers(0, 256, (4, 42, 42, 4)).astype(np.uint8)
    elif which == "ppo_mlp":
        spec, ospec = netspec.ppo_mlp((4,), 2, (64, 64), "tanh", False), nets.ppo_mlp_spec((4,), 2, (64, 64), "tanh", False)
        obs = rng.standard_normal((7, 4)).astype(np.float32)
    elif which == "pendulum":
        spec = netspec.ppo_mlp((3,), 1, (64, 64), "tanh", False, action_type="DiagGaussian")
        ospec = nets.ppo_mlp_spec((3,), 1, (64, 64), "tanh", False, action_type="DiagGaussian")
        obs = rng.standard_normal((6, 3)).astype(np.float32)
    else:
        dim, a, mean, std = (84, 4, 0.0, 255.0) if which == "impala84" else (42, 6, 128.0, 128.0)
        spec, ospec = netspec.impala_cnn_opt((dim, dim, 4), a, mean, std), nets.impala_cnn_opt_spec((dim, dim, 4), a, mean, std)
        obs = rng.integers(0, 256, (3, dim, dim, 4)).astype(np.uint8)
    net = CpuActorCritic(spec, seed=0)
    params = nets.init_params(ospec, seed=3, bias_scale=0.05)
    net.set_weights({k: v.reshape(spec.names[k][1]) for k, v in params.items()})
    logits, value = net.forward(obs)
    ol, ov = nets.ActorCritic(ospec, params, np.float64).forward(obs)
    rel = lambda a, b: np.linalg.norm(np.asarray(a, np.float64) - b) / (np.linalg.norm(b) + 1e-30)
    assert logits.dtype == np.float32 and logits.shape == ol.shape and value.shape == (len(obs),)
    assert rel(logits, ol) < 1e-5 and rel(value, ov[:, 0]) < 1e-5, (rel(logits, ol), rel(value, ov[:, 0]))
    got = net.get_weights()
    assert list(got) == list(spec.names) and all(np.array_equal(got[k].reshape(-1), params[k].reshape(-1).astype(np.float32))
                                                  for k in params)
    with pytest.raises(KeyError):
        net.set_weights({"no_such_variable": np.zeros(3)})
    net.set_weights({"no_such_variable": np.zeros(3), spec.pi_name + "/bias": np.ones(spec.action_dim, np.float32)})
    assert np.array_equal(net.get_weights()[spec.pi_name + "/bias"], np.ones(spec.action_dim, np.float32))


def test_explorer_side_models_predict_on_cpu_with_learner_weights(tmp_path):
    """Every registered model class builds as the CPU replica in a GPU-less process (explorer.py:60), accepts a
    weights dict / .npz published by a learner of the same configuration and returns ``predict`` in the reference's
    shapes (ppo.py:104-109, impala_cnn_opt.py:267-277, impala_cnn.py predict)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present: explorer processes run with CUDA_VISIBLE_DEVICES=-1")
    from xingtian_amd.model import model_builder
    m = model_builder({"model_name": "PpoCnn", "state_dim": [84, 84, 4], "action_dim": 4, "input_dtype": "uint8",
                       "model_config": {"VF_SHARE_LAYERS": True, "hidden_sizes": [256], "activation": "relu", "SEED": 1}})
    a, lp, v = m.predict(np.zeros((2, 84, 84, 4), np.uint8))
    assert a.shape == (2,) and lp.shape == (2, 1) and v.shape == (2, 1) and lp.dtype == v.dtype == np.float32
    w = m.get_weights()
    w["pi_latent/bias"] = np.array([50.0, 0.0, 0.0, 0.0], np.float32)
    m.set_weights(w)
    a, lp, _ = m.predict(np.zeros((64, 84, 84, 4), np.uint8))
    assert (a == 0).all() and np.allclose(lp, 0.0, atol=1e-6)
    path = m.save_model(os.path.join(str(tmp_path), "actor_00001"))
    m2 = model_builder({"model_name": "PpoCnn", "state_dim": [84, 84, 4], "action_dim": 4, "input_dtype": "uint8",
                        "model_config": {"VF_SHARE_LAYERS": True, "hidden_sizes": [256], "activation": "relu", "SEED": 2}})
    m2.load_model(path)
    assert all(np.array_equal(m2.get_weights()[k], w[k]) for k in w)
    imp = model_builder({"model_name": "ImpalaCnnOpt", "state_dim": [42, 42, 4], "action_dim": 6, "input_dtype": "uint8",
                         "state_mean": 128.0, "state_std": 128.0, "model_config": {"sample_batch_step": 50, "SEED": 1}})
    logits, baseline, action = imp.predict(np.zeros((5, 42, 42, 4), np.uint8))
    assert logits.shape == (5, 6) and baseline.shape == (5,) and action.shape == (5,) and action.dtype == np.int32
    with pytest.raises(RuntimeError, match="inference-only"):
        imp.train(np.zeros((50, 42, 42, 4), np.uint8), [np.zeros((50, 6), np.float32), np.zeros(50, np.int32),
                                                        np.zeros(50, bool), np.zeros(50, np.float32)])
    ker = model_builder({"model_name": "ImpalaMlp", "state_dim": [6], "action_dim": 3, "model_config": {"SEED": 3}})
    p, val = ker.predict([np.zeros((4, 6), np.float32), np.zeros((4, 1), np.float32)])
    assert p.shape == (4, 3) and val.shape == (4, 1) and np.allclose(p.sum(-1), 1.0, atol=1e-6)


def test_product_never_imports_oracle():
    import subprocess
    import sys
    code = ("import sys; import xingtian_amd, xingtian_amd.model, xingtian_amd.algorithm, xingtian_amd.ops, "
            "xingtian_amd.parallel; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "xingtian_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_registry_and_import_config_semantics():
    from xingtian_amd.register import Registers, import_config
    import xingtian_amd.model  # noqa: F401
    import xingtian_amd.algorithm  # noqa: F401
    for name in ("PpoCnn", "PpoMlp", "ImpalaCnnOpt"):
        assert name in Registers.model
    for name in ("PPO", "IMPALAOpt"):
        assert name in Registers.algorithm
    with pytest.raises(KeyError):
        Registers.model["NoSuchModel"]
    g = {"LR": 1.0, "BATCH_SIZE": 2, "lower": 3}
    import_config(g, {"LR": 0.5, "UNKNOWN": 9, "lower": 4})
    assert g == {"LR": 0.5, "BATCH_SIZE": 2, "lower": 4}
    import_config(g, None)
    with pytest.raises(RuntimeError):
        Registers()


def test_yaml_surface_of_reference_examples(tmp_path):
    """the model_para blocks of the reference's example YAMLs parse into our specs (values copied from
    examples/breakout_ppo.yaml, cartpole_ppo.yaml, breakout_impala.yaml, pong_impala_speedup.yaml)."""
    import yaml
    from xingtian_amd.model import netspec
    cfg = yaml.safe_load("""
model_para:
  actor:
    model_name: PpoCnn
    state_dim: [84, 84, 4]
    action_dim: 4
    input_dtype: uint8
    model_config: {BATCH_SIZE: 320, NUM_SGD_ITER: 4, VF_SHARE_LAYERS: True, activation: relu, hidden_sizes: [256]}
""")["model_para"]["actor"]
    s = netspec.ppo_cnn(tuple(cfg["state_dim"]), cfg["action_dim"], tuple(cfg["model_config"]["hidden_sizes"]),
                        cfg["model_config"]["activation"], cfg["model_config"]["VF_SHARE_LAYERS"], cfg["input_dtype"])
    assert s.n_params == 847493 and s.input_xform == (1, 0.0, 255.0)


@pytest.mark.parametrize("which", ["ppo_cnn84", "ppo_cnn42_unshared", "ppo_mlp", "impala84", "impala42", "keras_cnn84",
                                   "keras_mlp"])
def test_netspec_matches_oracle_spec(which):
    from xingtian_amd.model import netspec
    if which == "ppo_cnn84":
        s, o = netspec.ppo_cnn((84, 84, 4), 4, (256,)), nets.ppo_cnn_spec((84, 84, 4), 4, (256,))
    elif which == "ppo_cnn42_unshared":
        s = netspec.ppo_cnn((42, 42, 4), 6, (64,), "tanh", False)
        o = nets.ppo_cnn_spec((42, 42, 4), 6, (64,), "tanh", False)
    elif which == "ppo_mlp":
        s, o = netspec.ppo_mlp((4,), 2), nets.ppo_mlp_spec((4,), 2)
    elif which == "keras_cnn84":      # non-opt ImpalaCnn: 32/64/64 convs + Dense 256 (impala_cnn.py:44-57)
        s, o = netspec.impala_cnn((84, 84, 4), 6), nets.impala_cnn_spec((84, 84, 4), 6)
        assert s.n_params == 8 * 8 * 4 * 32 + 32 + 4 * 4 * 32 * 64 + 64 + 3 * 3 * 64 * 64 + 64 + 3136 * 256 + 256 + 256 * 6 + 6 + 257
    elif which == "keras_mlp":
        s, o = netspec.impala_mlp((8,), 3, 128, 2), nets.impala_mlp_spec((8,), 3, 128, 2)
    elif which == "impala84":
        s, o = netspec.impala_cnn_opt((84, 84, 4), 4), nets.impala_cnn_opt_spec((84, 84, 4), 4)
    else:
        s = netspec.impala_cnn_opt((42, 42, 4), 6, 128.0, 128.0)
        o = nets.impala_cnn_opt_spec((42, 42, 4), 6, 128.0, 128.0)
    op = nets.init_params(o)
    assert set(op.keys()) == set(s.names.keys())
    assert s.n_params == sum(v.size for v in op.values())
    olayers = [l for tr in o["trunks"] for l in tr]
    assert len(olayers) == len(s.layers)
    for a, b in zip(s.layers, olayers):
        assert (a.OH, a.OW, a.PT, a.PL, a.N) == (b.out_h, b.out_w, b.pt, b.pl, b.cout), a.name
        # a full-image VALID conv is lowered to a dense layer over the flattened input (same K, same memory)
        assert a.K == (b.k * b.k * b.cin if b.kind == "conv" else b.cin), a.name
        assert a.param_off % 4 == 0
    # flat layout: blocks do not overlap
    spans = sorted((off, off + int(np.prod(shape))) for off, shape in s.names.values())
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0
    assert spans[-1][1] <= s.n_flat


def test_shard_helpers():
    from xingtian_amd import parallel
    for n, w in [(32, 8), (10, 4), (3, 8), (256, 8)]:
        got = [parallel.shard_range(n, r, w) for r in range(w)]
        assert got[0][0] == 0 and got[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
        sizes = [e - b for b, e in got]
        assert max(sizes) - min(sizes) <= 1
    perm = np.random.default_rng(0).permutation(100)
    parts = [parallel.split_minibatch(perm, 40, 40, r, 2) for r in range(2)]
    assert np.array_equal(np.concatenate(parts), perm[40:80])
    assert parallel.grad_scale("mean", 8) == 0.125 and parallel.grad_scale("sum", 8) == 1.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_worker(rank, world, port, out):
    """each rank: oracle gradients of ITS shard of a global minibatch (fp64, CPU) -> gloo all-reduce through
    xingtian_amd.parallel -> clip+Adam; rank 0 compares with the single-process oracle on the full minibatch."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xingtian_amd import parallel
    try:
        rng = np.random.default_rng(0)          # identical data on every rank
        spec = nets.ppo_cnn_spec((15, 15, 4), 3, (8,), "relu", True)
        params = nets.init_params(spec, seed=1, bias_scale=0.1)
        b = 12
        obs = rng.integers(0, 256, (b, 15, 15, 4)).astype(np.uint8)
        action = rng.integers(0, 3, b).astype(np.int32)
        old_logp = -np.abs(rng.standard_normal((b, 1))) - 0.5
        adv = rng.standard_normal((b, 1)); old_v = rng.standard_normal((b, 1))
        target_v = old_v + rng.standard_normal((b, 1))
        cfg = dict(LR=1e-3, LOSS_CLIPPING=0.2, ENTROPY_LOSS=0.01, VF_CLIP=1.0, CRITIC_LOSS_COEF=0.5, MAX_GRAD_NORM=0.5,
                   BATCH_SIZE=b, NUM_SGD_ITER=1)
        perm = rng.permutation(b)
        mine = parallel.split_minibatch(perm, 0, b, rank, world)
        assert parallel.world_info() == (rank, world)
        # local gradient of the GLOBAL-mean loss restricted to my rows = (|mine|/b) * grad of my local mean
        orc = nets.PpoLearnerOracle(spec, params, dict(cfg, BATCH_SIZE=len(mine)), np.float64)
        out_l = orc.step(obs[mine], action[mine], old_logp[mine], adv[mine], old_v[mine], target_v[mine], apply=False)
        flat = torch.from_numpy(np.concatenate([g.ravel() for g in out_l["grads"].values()]))
        parallel.allreduce_sum_(flat)                       # SUM over ranks
        flat *= parallel.grad_scale("mean", world)          # equal shards: mean of local means = global mean
        ref = nets.PpoLearnerOracle(spec, params, cfg, np.float64).step(
            obs[perm], action[perm], old_logp[perm], adv[perm], old_v[perm], target_v[perm], apply=False)
        ref_flat = np.concatenate([g.ravel() for g in ref["grads"].values()])
        np.testing.assert_allclose(flat.numpy(), ref_flat, rtol=1e-9, atol=1e-13)
        # replicas stay identical: the reduced buffer is bit-identical on every rank
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered)
        w = torch.tensor([float(rank + 1)])
        parallel.broadcast_weights_(w, src=0)
        assert w.item() == 1.0
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_data_parallel_gradient_allreduce_gloo_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: 1, 1: 1}


def _dp_impala_worker(rank, world, port, out):
    """IMPALA's loss is a SUM over trajectories (impala_cnn_opt.py:299-351): every rank owns whole trajectories, the
    all-reduced gradient needs NO 1/world scaling (parallel.grad_scale("sum")), and equals the single-process gradient
    of all trajectories."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xingtian_amd import parallel
    try:
        rng = np.random.default_rng(3)          # identical data on every rank
        spec = nets.impala_cnn_opt_spec((42, 42, 2), 3)
        params = nets.init_params(spec, seed=4, bias_scale=0.1)
        t, ntraj = 5, 6
        n = t * ntraj                            # env-major flat batch [traj * T + step]
        states = rng.integers(0, 256, (n, 42, 42, 2)).astype(np.uint8)
        bp = rng.standard_normal((n, 3)).astype(np.float32)
        act = rng.integers(0, 3, n).astype(np.int32)
        dones = rng.random(n) < 0.15
        rew = rng.choice([-1.0, 0.0, 1.0], n).astype(np.float32)
        cfg = dict(LR=1e-3, grad_norm_clip=40.0, sample_batch_step=t)
        lo, hi = parallel.shard_range(ntraj, rank, world)
        rows = slice(lo * t, hi * t)             # whole trajectories
        orc = nets.ImpalaLearnerOracle(spec, params, cfg, np.float64)
        mine = orc.step(states[rows], bp[rows], act[rows], dones[rows], rew[rows], apply=False)
        flat = torch.from_numpy(np.concatenate([g.ravel() for g in mine["grads"].values()]))
        parallel.allreduce_sum_(flat)
        flat *= parallel.grad_scale("sum", world)
        ref = nets.ImpalaLearnerOracle(spec, params, cfg, np.float64).step(states, bp, act, dones, rew, apply=False)
        ref_flat = np.concatenate([g.ravel() for g in ref["grads"].values()])
        np.testing.assert_allclose(flat.numpy(), ref_flat, rtol=1e-9, atol=1e-12)
        loss = torch.tensor([float(mine["loss"])], dtype=torch.float64)
        dist.all_reduce(loss)
        assert abs(loss.item() - float(ref["loss"])) < 1e-9 * max(1.0, abs(float(ref["loss"])))
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_data_parallel_impala_sum_loss_gloo_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_impala_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: 1, 1: 1}


def test_linear_cosine_decay_known_answers():
    """tf.train.linear_cosine_decay as the reference's ImpalaCnnOpt._get_lr uses it (impala_cnn_opt.py:236-249):
    closed-form values at 0, decay_steps/2, decay_steps and beyond."""
    import importlib
    mod = importlib.import_module("xingtian_amd.model.impala.impala_cnn_opt")
    lcd = mod.linear_cosine_decay
    lr0, ds = 0.01, 20000.0
    beta = 1e-6 / ds
    assert abs(lcd(lr0, 0, ds, beta=beta) - lr0 * (1.0 + beta)) < 1e-9
    assert abs(lcd(lr0, 10000, ds, beta=beta) - lr0 * (0.5 * 0.5 + beta)) < 1e-8      # linear 0.5, cos(pi/2) -> 0.5
    assert abs(lcd(lr0, 20000, ds, beta=beta) - lr0 * beta) < 1e-9
    assert lcd(lr0, 50000, ds, beta=beta) == lcd(lr0, 20000, ds, beta=beta)            # clamped at decay_steps
    assert lcd(lr0, 5000, ds, beta=beta).dtype == np.float32
    vals = [float(lcd(lr0, s_, ds, beta=beta)) for s_ in range(0, 20001, 500)]
    assert all(a > b for a, b in zip(vals, vals[1:]))                                   # monotone decreasing


@pytest.mark.skipif(not os.path.isfile("/root/reference/zeus/common/util/register.py"),
                    reason="the reference checkout only exists in the authoring container")
def test_reference_registry_accepts_the_hip_plugins_as_integration_md_says():
    """INTEGRATION.md section 1 against the reference's REAL registry: zeus/common/util/register.py is loaded
    unmodified (its two imports that need uninstalled packages are stubbed: `absl.logging` -> stdlib logging and
    `zeus.set_backend`, which would import TensorFlow, -> no-op) and the drop-in subclasses are
    registered and resolved by class __name__, which is how alg_para.alg_name / model_name select plugins
    (register.py:58-69)."""
    import importlib.util
    import logging as pylogging
    import types
    absl = types.ModuleType("absl")
    absl.logging = pylogging
    zeus = types.ModuleType("zeus")
    zeus.set_backend = lambda **kw: None
    saved = {k: sys.modules.get(k) for k in ("absl", "absl.logging", "zeus")}
    sys.modules["absl"], sys.modules["absl.logging"], sys.modules["zeus"] = absl, pylogging, zeus
    try:
        spec = importlib.util.spec_from_file_location("_ref_register", "/root/reference/zeus/common/util/register.py")
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    from xingtian_amd.algorithm.impala.impala_opt import IMPALAOpt as _IMPALAOpt
    from xingtian_amd.algorithm.ppo.ppo import PPO as _PPO
    from xingtian_amd.model.impala.impala_cnn_opt import ImpalaCnnOpt as _ImpalaCnnOpt
    from xingtian_amd.model.ppo.ppo_cnn import PpoCnn as _PpoCnn

    @ref.Registers.model
    class PpoCnnHip(_PpoCnn):
        pass

    @ref.Registers.model
    class ImpalaCnnOptHip(_ImpalaCnnOpt):
        pass

    @ref.Registers.algorithm
    class PPOHip(_PPO):
        pass

    @ref.Registers.algorithm
    class IMPALAOptHip(_IMPALAOpt):
        pass

    assert ref.Registers.model["PpoCnnHip"] is PpoCnnHip and ref.Registers.model["ImpalaCnnOptHip"] is ImpalaCnnOptHip
    assert ref.Registers.algorithm["PPOHip"] is PPOHip and ref.Registers.algorithm["IMPALAOptHip"] is IMPALAOptHip
    # same constructor contract as the reference's plugin classes: Algorithm(model_info, alg_config, **kw), Model(model_info)
    import inspect
    assert list(inspect.signature(PPOHip.__init__).parameters)[:3] == ["self", "model_info", "alg_config"]
    assert list(inspect.signature(PpoCnnHip.__init__).parameters)[:2] == ["self", "model_info"]


def _recording_model():
    from xingtian_amd.register import Registers

    class RecordingModel(object):
        def __init__(self, model_info):
            self.model_info = model_info
            self.calls = []

        def train(self, state, label, **kw):
            cp = lambda a: [np.array(x, copy=True) for x in a] if isinstance(a, (list, tuple)) else np.array(a, copy=True)
            self.calls.append((cp(state), cp(label)))
            return 0.5 + len(self.calls)

    Registers.model(RecordingModel)
    return RecordingModel


def _check_calls_against_golden(calls, z):
    assert len(calls) == int(z["train_ncalls"])
    for i, (state, label) in enumerate(calls):
        assert isinstance(state, list) == bool(z["train_%d_state_is_list" % i])
        st = state if isinstance(state, list) else [state]
        assert len(st) == int(z["train_%d_nstate" % i]) and len(label) == int(z["train_%d_nlabel" % i])
        for j, a in enumerate(st):
            ref = z["train_%d_state_%d" % (i, j)]
            assert a.dtype == ref.dtype and a.shape == ref.shape and np.array_equal(a, ref), ("state", i, j)
        for j, a in enumerate(label):
            ref = z["train_%d_label_%d" % (i, j)]
            assert a.dtype == ref.dtype and a.shape == ref.shape and np.array_equal(a, ref), ("label", i, j)


def test_algorithm_host_logic_matches_reference_executed_goldens(golden_dir):
    """What PPO.train / IMPALAOpt.train hand to Model.train -- concatenation order of ragged trajectories, dtypes,
    sequential BATCH_SIZE chunks (8, 8, 4 of 20 frames), the returned loss (mean of chunk losses), the cleared
    accumulators -- against fixtures produced by EXECUTING the reference's own classes with a recording model
    (oracle/gen_golden_alg.py; xt/algorithm/ppo/ppo.py:66-93, xt/algorithm/impala/impala_opt.py:73-147)."""
    from oracle import gen_golden_alg as G
    from xingtian_amd.algorithm import alg_builder
    _recording_model()
    z = np.load(os.path.join(golden_dir, "alg_ppo.npz"))
    alg = alg_builder("PPO", *G.PPO_CFG)
    assert alg.async_flag is False and alg.prepare_data_times == 3
    for tr in G.ppo_inputs():
        alg.prepare_data(tr)
    loss = alg.train()
    assert float(loss) == float(z["loss"])
    _check_calls_against_golden(alg.actor.calls, z)
    assert (alg.obs == [] and alg.adv == []) == bool(z["lists_cleared"])
    z = np.load(os.path.join(golden_dir, "alg_impala_opt.npz"))
    alg = alg_builder("IMPALAOpt", *G.IMPALA_CFG)
    assert alg.async_flag is False and alg.prepare_data_times == 2
    for m in G.impala_inputs():
        alg.prepare_data(m)
    loss = alg.train()
    assert float(loss) == float(z["loss"])
    _check_calls_against_golden(alg.actor.calls, z)
    assert (alg.states == [] and alg.rewards == []) == bool(z["lists_cleared"])


def test_plain_impala_host_vtrace_matches_reference_executed_goldens(golden_dir):
    """The non-opt IMPALA (xt/algorithm/impala/impala.py:31-190): actor forward over every stored state, numpy v-trace
    on PROBABILITIES with the reference's own index convention, (state, pg_adv) / (one-hot, target) in sequential
    BATCH_SIZE chunks of 5, 5, 2 -- against what the EXECUTED reference class handed to its model for the same
    fragments and the same model outputs (oracle/gen_golden_alg.py)."""
    from oracle import gen_golden_alg as G
    from xingtian_amd.algorithm import alg_builder
    from xingtian_amd.algorithm.impala.impala import vtrace_from_probs
    z = np.load(os.path.join(golden_dir, "alg_impala.npz"))
    rec = _recording_model()

    def predict(self, state):
        if len(state[0]) == 1:
            return [z["single_pred_p"], z["single_pred_v"]]
        assert np.array_equal(state[0], z["pred_state"]) and state[1].shape == (len(state[0]), 1)
        return [z["pred_p"], z["pred_v"]]

    rec.predict = predict
    alg = alg_builder("IMPALA", *G.IMPALA_PLAIN_CFG)
    assert alg.async_flag is False and alg.episode_len == 6 and alg.prepare_data_times == 2
    for m in G.impala_plain_inputs():
        alg.prepare_data(m)
    loss = alg.train()
    assert float(loss) == float(z["loss"])
    _check_calls_against_golden(alg.actor.calls, z)
    assert (alg.state == [] and alg.rewards == []) == bool(z["lists_cleared"])
    single = alg.predict(np.zeros((6, 6, 2), np.uint8))
    assert np.array_equal(single[0], z["single_pred_p"]) and np.array_equal(single[1], z["single_pred_v"])
    # the recursion itself on a hand-checkable case: one fragment, T = 2, rho = 1, no terminal
    p = np.full((1, 2, 2), 0.5)
    onehot = np.array([[[1.0, 0.0], [0.0, 1.0]]])
    pg, tgt = vtrace_from_probs(p, p, onehot, np.ones((1, 2, 1)), np.zeros((1, 2, 1), bool), np.zeros((1, 2, 1)),
                                np.zeros((1, 2, 1)), 0.5)
    assert np.allclose(tgt[0, :, 0], [1.5, 1.0]) and np.allclose(pg[0, :, 0], [1.5, 1.0])


def test_architecture_tables_match_reference_executed_goldens(golden_dir):
    """get_default_filters / get_atari_filter / default hidden sizes + activations and the INFERRED architecture of
    table-less square observations, against values produced by executing the reference (oracle/gen_golden_arch.py;
    xt/model/model_utils.py:100-176, xt/model/atari_model.py:4-23) -- for the HIP netspec and for the oracle."""
    import json
    from xingtian_amd.model import netspec
    g = json.load(open(os.path.join(golden_dir, "arch_tables.json")))
    for key, ref in g["ppo_cnn_filters"].items():
        h = int(key.split("x")[0])
        assert [list(f) for f in netspec.ppo_cnn_filters((h, h, 4))] == ref
        assert [list(f) for f in nets.ppo_cnn_filters((h, h, 4))] == ref
    for key, ref in g["impala_filters"].items():
        h = int(key.split("x")[0])
        assert [list(f) for f in netspec.impala_filters((h, h, 4))] == ref
        assert [list(f) for f in nets.impala_filters((h, h, 4))] == ref
    for key, ref in g["ppo_cnn_inferred"].items():
        h = int(key.split("x")[0])
        if h <= 64:
            assert [list(f) for f in netspec.ppo_cnn_filters((h, h, 4))] == ref, key
            assert [list(f) for f in nets.ppo_cnn_filters((h, h, 4))] == ref, key
        else:   # the reference's rule gives a kernel larger than the image there: we refuse instead
            assert ref[0][1] > h
            with pytest.raises(ValueError):
                netspec.ppo_cnn_filters((h, h, 4))
    assert g["ppo_cnn_rank2_raises"]
    with pytest.raises(ValueError):
        netspec.ppo_cnn_filters((84, 84))
    from xingtian_amd.model.ppo.ppo_cnn import PpoCnn
    from xingtian_amd.model.ppo.ppo_mlp import PpoMlp
    for cls, want in ((PpoCnn, g["cnn_defaults"]), (PpoMlp, g["mlp_defaults"])):
        probe = cls.__new__(cls)                       # option parsing only: no GPU, no network
        probe._read_trunk_options({}, None, *cls.TRUNK_DEFAULTS)
        assert probe.hidden_sizes == want["hidden_sizes"] and probe.activation == want["activation"]


def test_defaults_are_the_reference_module_constants(golden_dir):
    """xingtian_amd/defaults.py against tests/golden/defaults.json (oracle/gen_golden_arch.py executes the
    reference's four default_config files)."""
    import json
    from xingtian_amd import defaults
    with open(os.path.join(golden_dir, "defaults.json")) as f:
        ref = json.load(f)
    assert set(ref) == set(defaults.DEFAULTS)
    for key, vals in ref.items():
        assert defaults.DEFAULTS[key] == vals, key


def test_ppo_train_accepts_the_learner_threads_kwargs():
    """learner.py:348 calls ``alg.train(episode_num=...)``; only ``perms`` may reach the model."""
    from xingtian_amd.algorithm.ppo.ppo import PPO

    class _Actor(object):
        stream_ingest = False

        def train(self, state, label, perms=None):
            self.got = (state[0].shape, perms)
            return np.float32(0.5)

    alg = PPO.__new__(PPO)
    from xingtian_amd.algorithm.algorithm import RolloutFields
    alg._rollout, alg._streamed, alg.actor = RolloutFields(*PPO.FIELDS), 0, _Actor()
    t = 5
    alg.prepare_data({"cur_state": np.zeros((t, 4), np.float32), "action": np.zeros(t, np.int32),
                      "logp": np.zeros((t, 1), np.float32), "adv": np.zeros((t, 1)), "old_value": np.zeros((t, 1), np.float32),
                      "target_value": np.zeros((t, 1))})
    assert alg.train(episode_num=3) == np.float32(0.5)
    assert alg.actor.got == ((t, 4), None)


def test_ppo_cnn_netspec_rejects_unknown_input_dtype():
    from xingtian_amd.model import netspec
    with pytest.raises(ValueError):
        netspec.ppo_cnn((84, 84, 4), 4, (256,), "relu", True, input_dtype="uint16")


# ---------------------------------------------------------------------------------------------- transport (f1)
def _ppo_message(rng, t=16):
    return {"cur_state": rng.integers(0, 256, (t, 84, 84, 4)).astype(np.uint8), "action": rng.integers(0, 4, t).astype(np.int32),
            "logp": rng.standard_normal((t, 1)).astype(np.float32), "adv": rng.standard_normal((t, 1)),
            "old_value": rng.standard_normal((t, 1)).astype(np.float32), "target_value": rng.standard_normal((t, 1)),
            "reward": [float(x) for x in rng.choice([-1.0, 0.0, 1.0], t)], "done": [bool(x) for x in rng.random(t) < 0.1],
            "info": [{"real_done": bool(i % 3 == 0), "eval_reward": float(i)} for i in range(t)]}


def test_transport_codec_round_trips_the_reference_payloads():
    """The wire payloads of the learner path (SURVEY 8b ``train_data``; the weights dict of get_weights) through
    xingtian_amd.transport: dtypes, shapes, field order, python lists (reward / done / info) and numpy scalars survive;
    arrays come back as zero-copy views into the message buffer."""
    from collections import OrderedDict
    from xingtian_amd import transport
    rng = np.random.default_rng(0)
    msg = _ppo_message(rng)
    ctr = {"cmd": "train", "broker_id": 0, "explorer_id": 3, "agent_id": -1}
    buf = transport.encode(ctr, msg)
    ctr2, out = transport.decode(buf)
    assert ctr2 == ctr and list(out) == list(msg)
    for k, v in msg.items():
        if isinstance(v, np.ndarray):
            assert out[k].dtype == v.dtype and out[k].shape == v.shape and np.array_equal(out[k], v)
            assert not out[k].flags.owndata                      # a view into buf, not a copy
        else:
            assert out[k] == v
    imp = {"cur_state": rng.integers(0, 256, (250, 42, 42, 4)).astype(np.uint8), "logit": rng.standard_normal((250, 6)).astype(np.float32),
           "action": rng.integers(0, 6, 250).astype(np.int32), "reward": list(rng.choice([-1.0, 0.0, 1.0], 250)),
           "done": list(rng.random(250) < 0.1), "info": [{} for _ in range(250)]}
    _, out = transport.decode(transport.encode({"cmd": "train"}, imp))
    assert out["done"] == [bool(x) for x in imp["done"]] and out["reward"] == [float(x) for x in imp["reward"]]
    assert np.array_equal(out["logit"], imp["logit"])
    weights = OrderedDict((("shared_conv_layer_0/kernel", rng.standard_normal((8, 8, 4, 32)).astype(np.float32)),
                           ("shared_conv_layer_0/bias", np.zeros(32, np.float32)), ("pi_latent/kernel", np.zeros((256, 4), np.float32))))
    _, w2 = transport.decode(transport.encode({"cmd": "explore"}, weights))
    assert list(w2) == list(weights) and all(np.array_equal(w2[k], weights[k]) for k in weights)
    with pytest.raises(ValueError):
        transport.decode(b"nope" + bytes(60))
    seen = []
    transport.decode_into(buf, lambda data, ctr_info=None: seen.append((ctr_info["cmd"], data["cur_state"].sum())))
    assert seen == [("train", msg["cur_state"].sum())]


def _ring_producer(name, n_msgs, seed):
    from xingtian_amd import transport
    ring = transport.ShmRing(name=name, create=False, slots=4, slot_bytes=4 << 20)
    rng = np.random.default_rng(seed)
    for i in range(n_msgs):
        assert ring.send({"cmd": "train", "seq": i}, _ppo_message(rng), timeout=30.0)
    ring.close()


def test_shared_memory_ring_carries_rollouts_between_processes():
    """ShmRing (the plasma-free intra-node channel): a producer PROCESS pushes 12 trajectories of 16 x 84x84x4 frames
    through a 4-slot ring (it wraps three times and blocks when full); the consumer receives them in order, bit for
    bit, half of them zero-copy (recv_into) and half as copies (recv)."""
    from xingtian_amd import transport
    ring = transport.ShmRing(slots=4, slot_bytes=4 << 20)
    ctx = mp.get_context("spawn")
    proc = ctx.Process(target=_ring_producer, args=(ring.name, 12, 5))
    proc.start()
    try:
        rng = np.random.default_rng(5)
        for i in range(12):
            want = _ppo_message(rng)
            if i % 2:
                got = ring.recv(timeout=60.0)
                assert got is not None
                ctr, data = got
                assert data["cur_state"].flags.owndata or data["cur_state"].base is not None
            else:
                box = {}
                ctr = ring.recv_into(lambda d, ctr_info=None: box.update({k: (v.copy() if isinstance(v, np.ndarray) else v)
                                                                           for k, v in d.items()}), timeout=60.0)
                assert ctr is not None
                data = box
            assert ctr == {"cmd": "train", "seq": i}
            for k, v in want.items():
                assert np.array_equal(data[k], v) if isinstance(v, np.ndarray) else data[k] == v, (i, k)
        assert ring.pending() == 0 and ring.recv(block=False) is None
        proc.join(30)
        assert proc.exitcode == 0
    finally:
        if proc.is_alive():
            proc.terminate()
        ring.close()


def test_yaml_to_alg_para_matches_the_reference_executed_patching():
    """xingtian_amd.config (YAML -> alg_para -> alg_builder) against what the reference's OWN
    patch_alg_within_config / patch_model_config_by_env_info / setup_learner produce for its example YAMLs
    (tests/golden/learner_config.json, executed by oracle/gen_golden_cfg.py)."""
    import json
    from xingtian_amd import config as cfg
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "learner_config.json")))
    assert len(golden) == 15      # every PPO / IMPALA example of the reference (oracle/gen_golden_cfg.py)
    for rel, g in golden.items():
        got = cfg.learner_alg_para(g["config"], g["env_info"])
        assert got == g["alg_para"], rel
        assert got["model_info"]["actor"]["type"] == "learner"
        # the YAML without an explicit node_config means one local node
        bare = {k: v for k, v in g["config"].items() if k != "node_config"}
        assert cfg.learner_alg_para(bare, g["env_info"]) == g["alg_para"], rel
    if not torch.cuda.is_available():
        # explorer side of the same YAML: the CPU replica, with the reference's predict shapes
        g = golden["examples/breakout_ppo.yaml"]
        actor = cfg.build_explorer_model(g["config"], g["env_info"])
        assert actor.net.inference_only
        a, lp, v = actor.predict(np.zeros((2, 84, 84, 4), np.uint8))
        assert a.shape == (2,) and lp.shape == (2, 1) and v.shape == (2, 1)
        with pytest.raises(RuntimeError, match="no HIP device"):
            cfg.build_learner_algorithm(g["config"], g["env_info"])


def test_transport_codec_property_random_payloads():
    """hypothesis: arbitrary dicts of ndarrays (any of the wire dtypes, 0-d to 4-d, empty included) and python
    scalars / lists / nested dicts survive encode -> decode exactly, with the field order preserved."""
    from hypothesis import given, settings, strategies as st
    from hypothesis.extra import numpy as hnp
    from xingtian_amd import transport
    dtypes = st.sampled_from([np.uint8, np.int32, np.int64, np.float32, np.float64, np.bool_])
    arrays = dtypes.flatmap(lambda dt: hnp.arrays(dt, hnp.array_shapes(min_dims=0, max_dims=4, min_side=0, max_side=5)))
    scalars = st.one_of(st.integers(-2 ** 40, 2 ** 40), st.floats(allow_nan=False, allow_infinity=False), st.booleans(),
                        st.text(max_size=8), st.none())
    objects = st.one_of(scalars, st.lists(scalars, max_size=6), st.dictionaries(st.text(max_size=4), scalars, max_size=3))
    payloads = st.dictionaries(st.text(min_size=1, max_size=10), st.one_of(arrays, objects), max_size=6)

    @settings(max_examples=60, deadline=None, database=None)
    @given(payloads, st.dictionaries(st.text(min_size=1, max_size=6), scalars, max_size=4))
    def check(data, ctr):
        buf = transport.encode(ctr, data)
        ctr2, out = transport.decode(buf)
        assert ctr2 == ctr and list(out) == list(data)
        for k, v in data.items():
            if isinstance(v, np.ndarray):
                # bytes, not values: NaN payloads (and their bit patterns) must survive too
                assert out[k].dtype == v.dtype and out[k].shape == v.shape
                assert np.ascontiguousarray(out[k]).tobytes() == np.ascontiguousarray(v).tobytes()
            else:
                assert out[k] == v

    check()


def test_shard_partition_properties():
    """hypothesis: for any (items, world) the rank shards of ``shard_range`` / ``split_minibatch`` are contiguous,
    disjoint, cover everything exactly once and differ in size by at most one -- including short last minibatches
    and more ranks than rows."""
    from hypothesis import given, settings, strategies as st
    from xingtian_amd import parallel

    @settings(max_examples=200, deadline=None)
    @given(st.integers(0, 5000), st.integers(1, 64))
    def ranges(n, world):
        spans = [parallel.shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[r][1] == spans[r + 1][0] for r in range(world - 1))
        sizes = [e - b for b, e in spans]
        assert min(sizes) >= 0 and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)

    @settings(max_examples=100, deadline=None)
    @given(st.integers(1, 700), st.integers(1, 320), st.integers(1, 16), st.integers(0, 2 ** 31 - 1))
    def minibatches(n, bsz, world, seed):
        perm = np.random.default_rng(seed).permutation(n).astype(np.int32)
        seen = []
        for start in range(0, n, bsz):
            parts = [parallel.split_minibatch(perm, start, bsz, r, world) for r in range(world)]
            assert np.array_equal(np.concatenate(parts), perm[start:start + bsz])
            seen.append(np.concatenate(parts))
        assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(n))

    ranges()
    minibatches()


def test_native_staging_pool_copies_bit_for_bit_and_tunes_itself():
    """xt_stage_rows (the native worker pool that replaces the Python np.copyto into pinned staging): every thread
    count / store kind / chunk size / misalignment copies bit for bit and never writes outside its range; xt_stage_tune
    measures the ten variants on this host and keeps one of them (no GPU involved: dev_dst = NULL)."""
    import ctypes
    from xingtian_amd import ingest, lib
    h = lib.load()
    rep = ingest.staging_report()
    assert rep["threads"] in (0, 1, 2, 4, 8) and len(rep["gbps"]) == 10 and all(v > 0 for v in rep["gbps"].values())
    picked = (ctypes.c_int32(), ctypes.c_int32())
    h.xt_stage_get(ctypes.byref(picked[0]), ctypes.byref(picked[1]))
    rng = np.random.default_rng(3)
    try:
        for threads, nt in ((0, 0), (0, 1), (1, 1), (3, 0), (8, 1)):
            lib.check(h.xt_stage_set(threads, nt), "xt_stage_set")
            for n, chunk in ((1, 0), (63, 0), (64 * 1024 + 5, 4096), (3612672, 0), (3612672 + 17, 1 << 18)):
                src = rng.integers(0, 256, n, dtype=np.uint8)
                dst = np.full(n + 128, 7, np.uint8)
                for off in (0, 3, 64):
                    dst[:] = 7
                    lib.check(h.xt_stage_rows(dst[off:].ctypes.data, src.ctypes.data, n, None, chunk, 0, -1, None), "xt_stage_rows")
                    assert np.array_equal(dst[off:off + n], src), (threads, nt, n, off)
                    assert (dst[:off] == 7).all() and (dst[off + n:] == 7).all(), (threads, nt, n, off)
        assert h.xt_stage_set(99, 0) != 0 and b"threads" in h.xt_last_error()
        assert h.xt_stage_rows(None, None, 4, None, 0, 0, -1, None) != 0
    finally:
        h.xt_stage_set(picked[0].value, picked[1].value)


def test_rollout_fields_copy_transport_views_but_keep_owned_arrays():
    """with STREAM_INGEST off, prepare_data keeps the arriving arrays until train(); a zero-copy transport
    hands it views into a slot that is recycled right after the call, so views are copied and owned arrays are not."""
    from xingtian_amd import transport
    from xingtian_amd.algorithm.algorithm import RolloutFields
    msg = bytearray(transport.encode({"cmd": "train"}, {"cur_state": np.arange(24, dtype=np.uint8).reshape(2, 12),
                                                         "action": np.array([1, 2], np.int32)}))
    _, data = transport.decode(msg)
    owned = np.array([5, 6], np.int32)
    rf = RolloutFields("cur_state", "action")
    rf.add(cur_state=data["cur_state"], action=owned)
    for i in range(len(msg)):          # the producer overwrites the slot
        msg[i] = 0xff
    assert np.array_equal(rf.parts["cur_state"][0], np.arange(24, dtype=np.uint8).reshape(2, 12))
    assert rf.parts["action"][0] is owned


def test_data_parallel_guards_fail_identically_on_every_rank():
    """strict sharding of a minibatch with fewer rows than ranks must fail on EVERY rank before the first
    collective (only the empty ranks raised: the others hung in the all-reduce); the same holds for the hook-routed
    data-parallel IMPALA step."""
    from xingtian_amd import lib, parallel
    cfg = dict(BATCH_SIZE=8, NUM_SGD_ITER=1, LR=1e-3, MAX_GRAD_NORM=1.0)
    perm = torch.arange(8 + 3, dtype=torch.int32).reshape(1, -1)       # last minibatch: 3 rows
    for rank in range(4):
        class _Net(object):
            def make_ppo_cfg(self, *a, **k):
                return None

            def ppo_step(self, *a, **k):
                pass

            def apply(self, *a, **k):
                pass
            grads = torch.zeros(4)
            loss_out = None
        with pytest.raises(ValueError, match="3 rows cannot be split over 4 ranks"):
            parallel.dp_ppo_update(_Net(), cfg, None, perm, None, None, None, None, None, rank, 4, mode="strict")
    # (data-parallel IMPALA: both branches of dp_impala_step accept fewer trajectories than ranks since round 5 -- the empty
    # ranks contribute a zero gradient, tests/test_gpu_dp.py::test_data_parallel_impala_with_an_empty_shard)


def _fanin_producer(name, explorer_id, n_msgs, slots, slot_bytes):
    from xingtian_amd import transport
    ring = transport.RingSet.attach(name, slots=slots, slot_bytes=slot_bytes)
    rng = np.random.default_rng(1000 + explorer_id)
    for seq in range(n_msgs):
        obs = rng.integers(0, 256, (4, 84, 84, 4), dtype=np.uint8)
        ok = ring.send({"cmd": "train", "from": explorer_id, "seq": seq},
                       {"cur_state": obs, "action": np.full(4, explorer_id, np.int32), "sum": int(obs.sum())}, timeout=120.0)
        assert ok
    ring.close()


def test_ring_set_fans_32_explorer_processes_into_one_learner_with_back_pressure():
    """SURVEY 8 f1 fan-in: 32 producer PROCESSES, one 2-slot ring each (every producer has more messages than slots:
    it blocks until the learner drains), one consumer polling round robin.  Every message arrives exactly once, intact,
    in per-explorer order, tagged with the explorer it came from; a sweep serves at most one message per ring."""
    from xingtian_amd import transport
    n_exp, n_msgs, slots, slot_bytes = 32, 5, 2, 1 << 18
    rs = transport.RingSet(n_exp, slots=slots, slot_bytes=slot_bytes)
    ctx = mp.get_context("fork")
    procs = [ctx.Process(target=_fanin_producer, args=(rs.names[i], i, n_msgs, slots, slot_bytes)) for i in range(n_exp)]
    for p in procs:
        p.start()
    seen = {i: [] for i in range(n_exp)}
    sweeps = []

    def sink(data, ctr_info=None):
        i = ctr_info["explorer_id"]
        assert ctr_info["from"] == i and ctr_info["cmd"] == "train"
        assert not data["cur_state"].flags.owndata                      # zero-copy view into the slot
        assert int(data["cur_state"].sum()) == data["sum"] and (data["action"] == i).all()
        seen[i].append(ctr_info["seq"])

    try:
        import time as _time
        t0 = _time.monotonic()
        total = 0
        while total < n_exp * n_msgs and _time.monotonic() - t0 < 120.0:
            k = rs.poll_into(sink)
            assert k <= n_exp
            if k:
                sweeps.append(k)
            else:
                _time.sleep(0.001)
            total += k
        assert total == n_exp * n_msgs
        for i in range(n_exp):
            assert seen[i] == list(range(n_msgs)), (i, seen[i])
        assert rs.served == [n_msgs] * n_exp and rs.pending() == 0
        assert rs.recv_many_into(sink, 1, timeout=0.05) == 0
        for p in procs:
            p.join(30)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        rs.close()


def _weights_reader(name, slot_bytes, last_seq, out_q):
    from xingtian_amd import transport
    import time as _time
    ring = transport.WeightsRing(name=name, slot_bytes=slot_bytes, create=False)
    seqs, t0 = [], _time.monotonic()
    while (not seqs or seqs[-1] < last_seq) and _time.monotonic() - t0 < 120.0:
        got = ring.fetch()
        if got is None:
            _time.sleep(0.0005)
            continue
        seq, ctr, w = got
        ok = ctr["seq"] == seq and ctr["cmd"] == "weights" and all(v.flags.owndata and (v == float(seq)).all() for v in w.values()) \
            and list(w) == ["conv/kernel:0", "conv/bias:0", "dense/kernel:0"]
        if not ok:
            out_q.put(("torn", seq))
            return
        seqs.append(seq)
    out_q.put(("ok", seqs))
    ring.close()


def test_weights_ring_publishes_to_many_readers_without_torn_reads():
    """SURVEY 8 f1 fan-out: one writer publishes 60 versions of a 1.2 MB weights dict (every element = the version
    number, so a torn or half-written read is visible), four reader PROCESSES fetch concurrently: every fetch is one
    internally consistent publish, sequence numbers only grow, everybody ends on the last version."""
    from xingtian_amd import transport
    slot_bytes = 2 << 20
    ring = transport.WeightsRing(slot_bytes=slot_bytes, slots=3)
    assert ring.fetch() is None
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    last = 60
    readers = [ctx.Process(target=_weights_reader, args=(ring.name, slot_bytes, last, q)) for _ in range(4)]
    for p in readers:
        p.start()
    try:
        shapes = {"conv/kernel:0": (8, 8, 4, 32), "conv/bias:0": (32,), "dense/kernel:0": (1152, 256)}
        import time as _time
        for v in range(1, last + 1):
            assert ring.publish({k: np.full(s, float(v), np.float32) for k, s in shapes.items()}, {"train_count": v}) == v
            _time.sleep(0.002 if v % 7 else 0.0)
        results = [q.get(timeout=120) for _ in readers]
        for status, seqs in results:
            assert status == "ok", (status, seqs)
            assert seqs[-1] == last and all(b > a for a, b in zip(seqs, seqs[1:]))
        with pytest.raises(ValueError):
            ring.publish({"big": np.zeros(slot_bytes, np.uint8)})
        for p in readers:
            p.join(30)
            assert p.exitcode == 0
    finally:
        for p in readers:
            if p.is_alive():
                p.terminate()
        ring.close()


def test_ring_releases_slots_in_order_behind_a_deferred_copy():
    """SlotGuard protocol of a pinned ring, without a GPU (fake events): a sink that holds its slot for an asynchronous
    copy keeps the ring from recycling it, later messages -- even ones taken with the copying ``recv`` -- queue behind it
    (slots are released in arrival order), the producer sees back pressure meanwhile, and everything is released once
    the copy's event reports completion."""
    from xingtian_amd import transport

    class FakeEvent(object):
        def __init__(self):
            self.done = False

        def query(self):
            return self.done

        def synchronize(self):
            self.done = True

    ring = transport.ShmRing(slots=2, slot_bytes=1 << 16)
    ring.pinned = True                 # (no hipHostRegister here: only the guard protocol is under test)
    ev = FakeEvent()
    try:
        msg = lambda i: transport.encode({"seq": i}, {"x": np.full(8, i, np.int32)})
        assert ring.send_bytes(msg(0)) and ring.send_bytes(msg(1))
        assert not ring.send_bytes(msg(2), block=False)                     # full
        seen = []
        ring.recv_into(lambda d, ctr_info=None: (seen.append(int(d["x"][0])), ctr_info["_slot_guard"].hold(ev)))
        assert seen == [0] and ring.pending() == 1 and len(ring._held) == 1
        assert not ring.send_bytes(msg(2), block=False)                     # slot 0 still belongs to the pending copy
        ctr, data = ring.recv(block=False)                                  # message 1, copied out: queues behind message 0
        assert ctr == {"seq": 1} and ring.pending() == 0 and len(ring._held) == 2
        assert not ring.send_bytes(msg(2), block=False)
        assert ring.recv(block=False) is None                               # nothing new; reaping finds the copy pending
        ev.done = True
        assert ring.reap() == 0                                             # both slots released, in order
        assert ring.send_bytes(msg(2), block=False) and ring.send_bytes(msg(3), block=False)
        got = []
        for _ in range(2):
            ring.recv_into(lambda d, ctr_info=None: got.append(int(d["x"][0])))
        assert got == [2, 3] and not ring._held
    finally:
        ring.pinned = False
        ring.close()


def test_ring_set_reaps_pinned_rings_whose_slots_are_all_held():
    """RingSet + a pinned ring (fake events, no GPU): every delivered message holds its slot behind a deferred copy that
    completes 20 ms later.  ``pending()`` does not count held slots, so the poller must reap them itself -- otherwise
    the ring is skipped forever, the producer blocks on 'ring full' and ``recv_many_into`` spins (the round-3 advisor's
    reproduction: 2 of 6 delivered).  All six messages must arrive, in order, and every slot must come back."""
    import threading
    from xingtian_amd import transport

    class TimedEvent(object):
        def __init__(self, delay):
            self.t = time.monotonic() + delay

        def query(self):
            return time.monotonic() >= self.t

        def synchronize(self):
            while not self.query():
                time.sleep(0.001)

    rs = transport.RingSet(1, slots=2, slot_bytes=1 << 16)
    rs.rings[0].pinned = True
    sent = []

    def producer():
        prod = transport.RingSet.attach(rs.names[0], slots=2, slot_bytes=1 << 16)
        for i in range(6):
            sent.append(prod.send_bytes(transport.encode({"seq": i}, {"x": np.full(4, i, np.int32)}), block=True, timeout=5.0))
        prod.close()

    th = threading.Thread(target=producer)
    th.start()
    seen = []

    def sink(data, ctr_info=None):
        seen.append(int(data["x"][0]))
        ctr_info["_slot_guard"].hold(TimedEvent(0.02))

    try:
        got = rs.recv_many_into(sink, 6, timeout=10.0)
        th.join(10.0)
        assert got == 6 and seen == list(range(6)) and all(sent), (got, seen, sent)
        time.sleep(0.05)
        assert rs.rings[0].reap() == 0 and rs.pending() == 0
    finally:
        rs.rings[0].pinned = False
        rs.close()


def test_image_observations_with_three_channels_pad_to_four_exactly():
    """examples/ant_ppo.yaml / dog_ppo.yaml: PpoCnn on [84, 84, 3] uint8 (the reference's get_cnn_backbone takes any
    channel count, xt/model/model_utils.py:49-80).  The first layer is stored as a [8, 8, 4, 32] block whose fourth
    input-channel rows are zero; the TF-shaped [8, 8, 3, 32] kernel is a strided view of it (get / set by name), and the
    CPU replica's forward on 3-channel frames equals the float64 oracle on the unpadded network."""
    from xingtian_amd.model import netspec
    from xingtian_amd.model.cpu_net import CpuActorCritic
    spec = netspec.ppo_cnn((84, 84, 3), 4, (512,), "relu", True)
    lay0 = spec.layers[0]
    assert lay0.C == 4 and lay0.kernel_shape == (8, 8, 3, 32) and spec.obs_channels_padded == 4
    assert spec.names["shared_conv_layer_0/kernel"][1] == (8, 8, 3, 32)
    assert spec.store_shape["shared_conv_layer_0/kernel"] == (8, 8, 4, 32)
    assert netspec.ppo_cnn((84, 84, 4), 4, (512,), "relu", True).obs_channels_padded is None
    net = CpuActorCritic(spec, seed=3)
    w = net.get_weights()
    assert w["shared_conv_layer_0/kernel"].shape == (8, 8, 3, 32)
    off, size = spec.var_extent("shared_conv_layer_0/kernel")
    block = net.params[off:off + size].reshape(8, 8, 4, 32)
    assert np.array_equal(block[:, :, :3], w["shared_conv_layer_0/kernel"]) and not block[:, :, 3].any()
    rng = np.random.default_rng(0)
    k = rng.standard_normal((8, 8, 3, 32)).astype(np.float32)
    net.set_weights({"shared_conv_layer_0/kernel": k})
    assert np.array_equal(net.get_weights()["shared_conv_layer_0/kernel"], k) and not block[:, :, 3].any()
    with pytest.raises(KeyError, match="shape"):
        net.set_weights({"shared_conv_layer_0/kernel": np.zeros((8, 8, 4, 32), np.float32)})
    # forward parity with the unpadded float64 oracle
    ospec = nets.ppo_cnn_spec((84, 84, 3), 4, (512,), "relu", True)
    params = nets.init_params(ospec, seed=5, bias_scale=0.05)
    net.set_weights({n: v.reshape(spec.names[n][1]) for n, v in params.items()})
    obs = rng.integers(0, 256, (3, 84, 84, 3)).astype(np.uint8)
    logits, value = net.forward(obs)
    olog, oval = nets.ActorCritic(ospec, params, np.float64).forward(obs)
    assert np.allclose(logits, olog, rtol=2e-4, atol=2e-5) and np.allclose(value, np.asarray(oval).reshape(-1), rtol=2e-4, atol=2e-5)
    # a uint8 transform with a non-zero mean pads with the byte that maps to zero; a fractional mean is refused by the
    # learner (HipActorCritic.obs_fill_byte), the CPU replica pads after the transform
    spec_i = netspec.impala_cnn_opt((84, 84, 3), 4, 128.0, 128.0)
    assert spec_i.layers[0].C == 4 and spec_i.obs_channels_padded == 4


def _stage_in_child(q):
    import ctypes as ct
    from xingtian_amd import lib
    h = lib.load()
    src = np.arange(1 << 20, dtype=np.uint8)
    dst = np.zeros_like(src)
    rc = h.xt_stage_rows(ct.c_void_p(dst.ctypes.data), ct.c_void_p(src.ctypes.data), src.nbytes, None, 0, 0, 4, None)
    q.put((rc, bool(np.array_equal(src, dst))))


def test_staging_pool_survives_fork():
    """the native staging pool's worker threads do not exist in a forked child and its mutexes may be copied
    locked; the child must build its own pool instead of waiting for workers that are not there."""
    import ctypes as ct
    import multiprocessing as mp_
    from xingtian_amd import lib
    h = lib.load()
    src = np.arange(1 << 20, dtype=np.uint8)[::-1].copy()
    dst = np.zeros_like(src)
    assert h.xt_stage_rows(ct.c_void_p(dst.ctypes.data), ct.c_void_p(src.ctypes.data), src.nbytes, None, 0, 0, 4, None) == 0
    assert np.array_equal(src, dst)              # the parent's pool now has 4 live workers
    ctx = mp_.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=_stage_in_child, args=(q,))
    p.start()
    p.join(30)
    assert not p.is_alive(), "the forked child hung in xt_stage_rows"
    assert q.get(timeout=5) == (0, True)
    dst[:] = 0
    assert h.xt_stage_rows(ct.c_void_p(dst.ctypes.data), ct.c_void_p(src.ctypes.data), src.nbytes, None, 0, 0, 4, None) == 0
    assert np.array_equal(src, dst)


def test_weights_ring_packed_publish_round_trips_names_shapes_and_padded_kernels():
    """The packed weight message (flat float32 parameter buffer + name table, what the learner's publish DMA-copies into a
    pinned slot) decodes on the reader side into the name-keyed dict of ``get_weights()`` -- also for a channel-padded
    first layer, whose TF-shaped kernel is a strided sub-block of its storage.  CPU: published from a CPU replica."""
    from xingtian_amd import transport
    from xingtian_amd.model import netspec
    from xingtian_amd.model.cpu_net import CpuActorCritic
    for sd in ((42, 42, 4), (42, 42, 3)):
        spec = netspec.ppo_cnn(sd, 3, (32,), "relu", True)
        net = CpuActorCritic(spec, seed=4)
        want = net.get_weights()
        ring = transport.WeightsRing(slot_bytes=2 << 20, slots=3)
        reader = transport.WeightsRing(name=ring.name, slot_bytes=2 << 20, slots=3, create=False)
        try:
            for rep in range(4):                                  # laps the three slots
                net.params += 0.5
                want = net.get_weights()
                k = ring.publish_flat_host(net.params, spec, {"train_count": rep})
                seq, ctr, got = reader.fetch()
                assert seq == k == rep + 1 and ctr["train_count"] == rep and ctr["cmd"] == "weights" and ctr["seq"] == k
                assert list(got) == list(want)
                for name in want:
                    assert got[name].shape == want[name].shape and np.array_equal(got[name], want[name]), name
            assert reader.fetch() is None
            replica = CpuActorCritic(spec, seed=0)
            replica.set_weights(got)
            back = replica.get_weights()          # (alignment padding between the blocks is not part of any variable)
            assert all(np.array_equal(back[name], want[name]) for name in want)
            with pytest.raises(ValueError, match="flat buffer"):
                ring.publish_flat_host(net.params[:-4], spec)
        finally:
            reader.close()
            ring.close()


def test_prefetcher_stages_one_train_ahead_in_arrival_order_and_hands_out_tokens():
    """transport.Prefetcher (asynchronous algorithms): a thread owns the consumer end of the ring and hands every message to
    ``alg.stage_message`` as it arrives -- at most ONE train ahead of the learner (it waits for ``staged_generation`` before
    the first message of the next train), in arrival order; the learner's unchanged loop ``recv_into(alg.prepare_data)`` x
    prepare_data_times gets a token per message.  Reference loop: xt/framework/learner.py:306-348."""
    import time
    from xingtian_amd import transport

    class FakeAlg(object):
        prepare_data_times = 2

        def __init__(self):
            self.gen, self.staged, self.booked = 0, [], []

        def stage_message(self, data, ctr_info=None):
            assert isinstance(data["cur_state"], np.ndarray) and data["cur_state"].shape == (4, 3)
            self.staged.append(int(ctr_info["k"]))
            return int(data["cur_state"].shape[0])

        def staged_generation(self):
            return self.gen

        def prepare_data(self, data, ctr_info=None):
            assert data == {"_prefetched": 4}
            self.booked.append(int(ctr_info["k"]))

    ring = transport.ShmRing(slots=8, slot_bytes=1 << 12)
    try:
        for k in range(6):
            assert ring.send({"k": k}, {"cur_state": np.full((4, 3), k, np.float32)})
        alg = FakeAlg()
        pf = transport.Prefetcher(ring, alg)
        try:
            t0 = time.monotonic()
            while len(alg.staged) < 2 and time.monotonic() - t0 < 5:
                time.sleep(0.001)
            time.sleep(0.05)
            assert alg.staged == [0, 1]                       # train 0 staged, train 1 NOT before the learner took train 0 over
            for train in range(3):
                assert pf.recv_many_into(alg.prepare_data, 2, timeout=5) == 2
                alg.gen += 1                                   # (what RolloutIngest.finish does inside alg.train())
                pf.notify()
                t0 = time.monotonic()
                want = min(6, 2 * (train + 2))
                while len(alg.staged) < want and time.monotonic() - t0 < 5:
                    time.sleep(0.001)
                assert alg.staged == list(range(want))
            assert alg.booked == list(range(6))
            assert pf.recv_into(alg.prepare_data, block=False) is None
        finally:
            pf.close()
    finally:
        ring.close()

    class Broken(FakeAlg):
        def stage_message(self, data, ctr_info=None):
            raise ValueError("boom")

    ring = transport.ShmRing(slots=2, slot_bytes=1 << 12)
    try:
        ring.send({"k": 0}, {"cur_state": np.zeros((4, 3), np.float32)})
        pf = transport.Prefetcher(ring, Broken())
        with pytest.raises(RuntimeError, match="staging thread failed"):
            pf.recv_into(lambda d, ctr_info=None: None, timeout=5)
        pf.close()
    finally:
        ring.close()


def test_sender_side_list_packing_ships_per_step_lists_as_typed_arrays():
    """transport.encode(pack_lists=True): the done / reward lists an agent ships (xt/agent/*: per-step python scalars) arrive as
    the arrays ``np.asarray`` makes of them -- what ``IMPALAOpt._data_proc`` / ``PPO.prepare_data`` compute anyway --, short,
    nested and mixed lists stay python objects, and without the option nothing changes."""
    from xingtian_amd import transport
    rng = np.random.default_rng(3)
    data = {"cur_state": rng.integers(0, 256, (16, 4, 4, 4)).astype(np.uint8), "done": [bool(b) for b in rng.random(16) < 0.3],
            "reward": [float(r) for r in rng.choice([-1.0, 0.0, 1.0], 16)], "ints": list(range(16)), "few": [1.0, 2.0],
            "nested": [[1, 2]] * 8, "mixed": [1, "a"] * 8, "np_bools": list(rng.random(16) < 0.5)}
    ctr, out = transport.decode(transport.encode({"cmd": "train"}, data, pack_lists=True))
    assert ctr == {"cmd": "train"} and list(out) == list(data)
    for key, dtype in (("done", np.bool_), ("reward", np.float64), ("ints", np.int64), ("np_bools", np.bool_)):
        assert isinstance(out[key], np.ndarray) and out[key].dtype == dtype and np.array_equal(out[key], np.asarray(data[key]))
    for key in ("few", "nested", "mixed"):
        assert isinstance(out[key], list) and out[key] == [list(x) if isinstance(x, list) else x for x in data[key]]
    _ctr, plain = transport.decode(transport.encode({"cmd": "train"}, data))
    assert isinstance(plain["done"], list) and plain["done"] == data["done"] and plain["reward"] == data["reward"]
    # the algorithm-side view is the same either way
    from xingtian_amd.algorithm.impala.impala_opt import IMPALAOpt
    msg = dict(data, logit=np.zeros((16, 2), np.float32), action=np.zeros(16, np.int32))
    a = IMPALAOpt._data_proc(transport.decode(transport.encode({}, msg, pack_lists=True))[1])
    b = IMPALAOpt._data_proc(transport.decode(transport.encode({}, msg))[1])
    for x, y in zip(a, b):
        assert np.asarray(x).dtype == np.asarray(y).dtype and np.array_equal(x, y)


def test_inline_prefetcher_stages_on_the_learner_thread_one_train_ahead():
    """transport.Prefetcher(inline=True): no thread -- ``pump_once()`` (the hook the model calls between two looks at the loss
    while the device trains) stages one waiting message per call, at most one train ahead of ``staged_generation``, in arrival
    order; ``recv_into`` pumps itself when no token is staged (the starved learner); the default picks the inline form exactly
    when the algorithm says its model calls the hook (``stage_inline_capable``)."""
    from xingtian_amd import transport

    class FakeAlg(object):
        prepare_data_times = 2

        def __init__(self, capable):
            self.gen, self.staged, self.booked, self.hook, self.capable = 0, [], [], "unset", capable

        def stage_message(self, data, ctr_info=None):
            self.staged.append(int(ctr_info["k"]))
            return int(data["cur_state"].shape[0])

        def staged_generation(self):
            return self.gen

        def stage_inline_capable(self):
            return self.capable

        def stage_inline(self, hook):
            self.hook = hook

        def prepare_data(self, data, ctr_info=None):
            assert data == {"_prefetched": 4}
            self.booked.append(int(ctr_info["k"]))

    ring = transport.ShmRing(slots=8, slot_bytes=1 << 12)
    try:
        for k in range(5):
            assert ring.send({"k": k}, {"cur_state": np.full((4, 3), k, np.float32)})
        alg = FakeAlg(capable=True)
        pf = transport.Prefetcher(ring, alg)                  # default -> inline
        assert pf.inline and pf._thread is None and alg.hook == pf.pump_once and alg.staged == []
        assert pf.pump_once() and pf.pump_once() and alg.staged == [0, 1]
        assert not pf.pump_once() and alg.staged == [0, 1]   # train 1 NOT before the learner has taken train 0 over
        assert pf.recv_many_into(alg.prepare_data, 2, timeout=5) == 2 and alg.booked == [0, 1]
        alg.gen += 1                                          # (RolloutIngest.finish inside alg.train())
        assert alg.hook() and alg.staged == [0, 1, 2]         # the model's call while the device trains
        assert pf.recv_into(alg.prepare_data, timeout=5) == {"k": 2}
        assert pf.recv_into(alg.prepare_data, timeout=5) == {"k": 3} and alg.staged == [0, 1, 2, 3]     # starved: pumped itself
        alg.gen += 1
        assert pf.recv_into(alg.prepare_data, timeout=5) == {"k": 4}
        assert pf.recv_into(alg.prepare_data, block=False) is None and alg.booked == [0, 1, 2, 3, 4]
        pf.close()
        assert alg.hook is None                               # detached
        alg2 = FakeAlg(capable=False)
        pf2 = transport.Prefetcher(ring, alg2)                # default -> the thread
        assert not pf2.inline and pf2._thread is not None and alg2.hook == "unset"
        pf2.close()
    finally:
        ring.close()
