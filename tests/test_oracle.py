"""CPU tests of the oracle itself: golden vectors from the executed reference (GAE),
autograd cross-checks of the hand-derived TF-graph restatement, properties."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import nets, returns, torch_ref


def test_gae_matches_reference_goldens_bit_exact(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "gae_*.npz")))
    assert len(files) >= 10
    for f in files:
        g = np.load(f)
        adv, old_v, tgt = returns.gae(g["value"], g["reward"], g["done"], float(g["gamma"]), float(g["lam"]))
        assert adv.dtype == np.float64 and old_v.dtype == np.float32
        assert np.array_equal(adv, g["adv"]), f
        assert np.array_equal(old_v, g["old_value"]), f
        assert np.array_equal(tgt, g["target_value"]), f


def test_gae_all_done_is_one_step_td():
    rng = np.random.default_rng(0)
    v = rng.standard_normal((17, 1)).astype(np.float32)
    r = rng.standard_normal(16)
    adv, _, _ = returns.gae(v, r, np.ones(16, bool))
    assert np.array_equal(adv[:, 0], r - v[:-1, 0].astype(np.float64))


def _tiny_cnn_spec():
    return nets.ppo_cnn_spec((15, 15, 4), 5, hidden_sizes=(16,), act="relu", vf_share=True)


def _labels(rng, b, a):
    action = rng.integers(0, a, b).astype(np.int32)
    old_logp = (-np.abs(rng.standard_normal((b, 1))) - 0.5)
    adv = rng.standard_normal((b, 1))
    old_v = rng.standard_normal((b, 1))
    target_v = old_v + rng.standard_normal((b, 1)) * 3
    return action, old_logp, adv, old_v, target_v


@pytest.mark.parametrize("which", ["cnn15", "mlp", "cnn_unshared"])
def test_ppo_grads_match_torch_autograd_fp64(which):
    rng = np.random.default_rng(1)
    b = 6
    if which == "cnn15":
        spec = _tiny_cnn_spec()
        obs = rng.integers(0, 256, (b, 15, 15, 4)).astype(np.uint8)
    elif which == "cnn_unshared":
        spec = nets.ppo_cnn_spec((15, 15, 4), 3, hidden_sizes=(8,), act="tanh", vf_share=False)
        obs = rng.integers(0, 256, (b, 15, 15, 4)).astype(np.uint8)
    else:
        spec = nets.ppo_mlp_spec((4,), 2)
        obs = rng.standard_normal((b, 4)).astype(np.float32)
    params = nets.init_params(spec, seed=3, bias_scale=0.1)
    cfg = dict(LR=3e-4, LOSS_CLIPPING=0.2, ENTROPY_LOSS=0.01, VF_CLIP=0.7, CRITIC_LOSS_COEF=0.8,
               MAX_GRAD_NORM=0.5, BATCH_SIZE=b, NUM_SGD_ITER=1)
    lab = _labels(rng, b, spec["action_dim"])
    o = nets.PpoLearnerOracle(spec, params, cfg, np.float64)
    t = torch_ref.TorchPpoLearner(spec, params, cfg, torch.float64)
    for it in range(3):   # also checks Adam + clip through 3 steps
        out = o.step(obs, *lab)
        tl, tg, gn = t.step(obs, *lab)
        assert abs(out["loss"] - tl) < 1e-10 * max(1, abs(tl))
        for k in tg:
            np.testing.assert_allclose(out["grads"][k], tg[k].numpy(), rtol=1e-8, atol=1e-12, err_msg=k)
        assert abs(out["gnorm"] - gn) < 1e-9
        for k in tg:
            np.testing.assert_allclose(o.net.params[k], t.net.params[k].detach().numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("act", ["sigmoid", "softsign", "softplus", "leaky_relu", "elu", "selu", "swish", "gelu"])
def test_monotonic_activations_match_torch_autograd_fp64(act):
    """The other entries of the reference's ACTIVATION_MAP (xt/model/model_utils.py:8-20; swish and gelu -- the tanh form
    of xt/model/tf_utils.py:157-166 -- take their derivative at the stored pre-activation): the oracle's
    forward and its hand-derived backward (pre-activation recovered from the saved output, textbook derivative) against
    torch autograd through torch's own definitions of the same functions, on a conv + dense PPO network."""
    rng = np.random.default_rng(5)
    b = 5
    spec = nets.ppo_cnn_spec((15, 15, 4), 3, hidden_sizes=(12,), act=act, vf_share=False)
    obs = rng.integers(0, 256, (b, 15, 15, 4)).astype(np.uint8)
    params = nets.init_params(spec, seed=2, bias_scale=0.2)
    cfg = dict(LR=3e-4, LOSS_CLIPPING=0.2, ENTROPY_LOSS=0.01, VF_CLIP=0.7, CRITIC_LOSS_COEF=0.8,
               MAX_GRAD_NORM=0.5, BATCH_SIZE=b, NUM_SGD_ITER=1)
    lab = _labels(rng, b, spec["action_dim"])
    out = nets.PpoLearnerOracle(spec, params, cfg, np.float64).step(obs, *lab)
    tl, tg, gn = torch_ref.TorchPpoLearner(spec, params, cfg, torch.float64).step(obs, *lab)
    assert abs(out["loss"] - tl) < 1e-10 * max(1, abs(tl))
    for k in tg:
        np.testing.assert_allclose(out["grads"][k], tg[k].numpy(), rtol=1e-8, atol=1e-12, err_msg=k)
    z = np.linspace(-30.0, 30.0, 241)
    y = nets.act_fwd(z, act)
    zt = torch.tensor(z, requires_grad=True)
    yt = torch_ref._ACT[act](zt)
    yt.sum().backward()
    # (gelu: 1 + tanh(.) cancels for very negative arguments -- both sides lose the same digits, in different orders)
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-12, atol=1e-300 if act != "gelu" else 1e-14)
    keep = np.abs(y) < 1 - 1e-9 if act == "softsign" else (y > 1e-12 if act == "softplus" else np.ones_like(y, bool))
    np.testing.assert_allclose(nets.act_bwd(np.ones_like(z), y, act, z)[keep], zt.grad.numpy()[keep], rtol=1e-6, atol=1e-12)


def test_gauss_ppo_grads_match_torch_autograd_fp64():
    """DiagGaussian PPO (pendulum_ppo.yaml shape: state 3, action 1, tanh 64-64 unshared) + a 3-dim action case."""
    for sd, ad in (((3,), 1), ((5,), 3)):
        spec = nets.ppo_mlp_spec(sd, ad, (64, 64), "tanh", False, action_type="DiagGaussian")
        params = nets.init_params(spec, seed=4, dtype=np.float64, bias_scale=0.1)
        assert list(params)[-1] == "pi_logstd" and params["pi_logstd"].shape == (1, ad)
        params["pi_logstd"] = np.random.default_rng(5).standard_normal((1, ad)) * 0.3
        rng = np.random.default_rng(6)
        b = 48
        obs = rng.uniform(-1, 1, (b,) + sd).astype(np.float32)
        act = rng.standard_normal((b, ad))
        lab = [act, -np.abs(rng.standard_normal((b, 1))) - 0.5, rng.standard_normal((b, 1)),
               rng.standard_normal((b, 1)), rng.standard_normal((b, 1))]
        cfg = dict(LR=3e-3, LOSS_CLIPPING=0.2, ENTROPY_LOSS=0.01, VF_CLIP=0.7, CRITIC_LOSS_COEF=1.0,
                   MAX_GRAD_NORM=0.5, BATCH_SIZE=b, NUM_SGD_ITER=1)
        o = nets.PpoLearnerOracle(spec, params, cfg, np.float64)
        t = torch_ref.TorchPpoLearner(spec, params, cfg, torch.float64)
        for it in range(3):
            out = o.step(obs, *lab)
            tl, tg, gn = t.step(obs, *lab)
            assert abs(out["loss"] - tl) < 1e-10 * max(1, abs(tl))
            for k in tg:
                np.testing.assert_allclose(out["grads"][k], tg[k].numpy(), rtol=1e-8, atol=1e-12, err_msg=k)
            assert abs(out["gnorm"] - gn) < 1e-9
            for k in tg:
                np.testing.assert_allclose(o.net.params[k], t.net.params[k].detach().numpy(), rtol=1e-9, atol=1e-12)


def test_keras_impala_grads_and_optimizer_match_torch_autograd_fp64():
    """Non-opt IMPALA models (Keras ``impala_loss`` on softmax outputs + 0.5*mse, tf.keras Adam with per-tensor
    clipnorm and lr decay): hand-derived float64 gradients and updates vs an independent torch-autograd version,
    CNN (clipnorm low enough to bite, decay exaggerated) and MLP (plain Adam)."""
    for which in ("cnn", "mlp"):
        if which == "cnn":
            spec = nets.impala_cnn_spec((36, 36, 4), 5)
            obs_f = lambda rng, b: rng.integers(0, 256, (b, 36, 36, 4)).astype(np.uint8)
            kw = dict(lr=3e-4, ent_coef=0.01, clipnorm=0.1, decay=0.05)
        else:
            spec = nets.impala_mlp_spec((6,), 3, 128, 2)
            obs_f = lambda rng, b: rng.uniform(-1, 1, (b, 6)).astype(np.float32)
            kw = dict(lr=3e-4, ent_coef=0.01)
        params = nets.init_params(spec, seed=8, dtype=np.float64, bias_scale=0.1)
        rng = np.random.default_rng(9)
        b, a = 24, spec["action_dim"]
        obs = obs_f(rng, b)
        adv = rng.standard_normal((b, 1))
        onehot = np.eye(a)[rng.integers(0, a, b)]
        tv = rng.standard_normal((b, 1))
        o = nets.KerasImpalaOracle(spec, params, dtype=np.float64, **kw)
        t = torch_ref.TorchKerasImpalaLearner(spec, params, dtype=torch.float64, **kw)
        for it in range(3):
            out = o.step(obs, adv, onehot, tv)
            tl, tg = t.step(obs, adv, onehot, tv)
            assert abs(out["loss"] - tl) < 1e-11 * max(1, abs(tl))
            for k in tg:
                np.testing.assert_allclose(out["grads"][k], tg[k].numpy(), rtol=1e-8, atol=1e-13, err_msg=k)
            for k in tg:
                np.testing.assert_allclose(o.net.params[k], t.net.params[k].detach().numpy(), rtol=1e-9, atol=1e-13)
        if which == "cnn":      # the clip must have been active on at least one tensor, inactive on another
            norms = [np.linalg.norm(g) for g in out["grads"].values()]
            assert max(norms) > kw["clipnorm"] > min(norms)


def test_same_padding_geometry():
    assert nets.conv_out_size(84, 8, 4, "same") == (21, 2, 2)
    assert nets.conv_out_size(21, 4, 2, "same") == (11, 1, 2)
    assert nets.conv_out_size(42, 4, 2, "same") == (21, 1, 1)
    assert nets.conv_out_size(84, 8, 4, "valid") == (20, 0, 0)


def test_param_counts_match_survey():
    p = nets.init_params(nets.ppo_cnn_spec((84, 84, 4), 4, hidden_sizes=(256,)))
    assert sum(v.size for v in p.values()) == 847493
    p = nets.init_params(nets.impala_cnn_opt_spec((84, 84, 4), 4))
    assert sum(v.size for v in p.values()) == 1005109
    p = nets.init_params(nets.impala_cnn_opt_spec((42, 42, 4), 6, 128.0, 128.0))
    assert sum(v.size for v in p.values()) == 1002551
    p = nets.init_params(nets.ppo_mlp_spec((4,), 2))
    assert sum(v.size for v in p.values()) == 9155


def test_impala_grads_match_torch_autograd_fp64():
    rng = np.random.default_rng(5)
    spec = nets.impala_cnn_opt_spec((42, 42, 4), 6, 128.0, 128.0)
    params = nets.init_params(spec, seed=2, bias_scale=0.05)
    tlen, bc = 5, 2
    n = tlen * bc
    obs = rng.integers(0, 256, (n, 42, 42, 4)).astype(np.uint8)
    bp = rng.standard_normal((n, 6)).astype(np.float32)
    act = rng.integers(0, 6, n).astype(np.int32)
    dones = rng.random(n) < 0.2
    rew = rng.standard_normal(n) * 2
    cfg = dict(LR=1e-3, grad_norm_clip=40.0, sample_batch_step=tlen, BATCH_SIZE=n)
    o = nets.ImpalaLearnerOracle(spec, params, cfg, np.float64)
    t = torch_ref.TorchImpalaLearner(spec, params, cfg, torch.float64)
    for it in range(2):
        out = o.step(obs, bp, act, dones, rew)
        tl, tg, gn = t.step(obs, bp, act, dones, rew)
        assert abs(out["loss"] - tl) < 1e-9 * max(1, abs(tl))
        for k in tg:
            np.testing.assert_allclose(out["grads"][k], tg[k].numpy(), rtol=1e-7, atol=1e-11, err_msg=k)


def test_vtrace_on_policy_equals_nstep_returns():
    """rho == 1 (same logits), no dones: vs_t = sum gamma^k r_{t+k} + gamma^n V_boot."""
    rng = np.random.default_rng(0)
    tl, b, a = 7, 3, 4
    lg = rng.standard_normal((tl, b, a))
    act = rng.integers(0, a, (tl, b))
    disc = np.full((tl, b), 0.9)
    rew = rng.standard_normal((tl, b))
    vals = rng.standard_normal((tl, b))
    boot = rng.standard_normal(b)
    vs, pg = returns.vtrace_from_logits(lg, lg, act, disc, rew, vals, boot)
    ret = boot.copy()
    for t in range(tl - 1, -1, -1):
        ret = rew[t] + 0.9 * ret
        np.testing.assert_allclose(vs[t], ret, rtol=1e-12)


def test_split_batches_env_major():
    x = np.arange(12)
    s = returns.split_batches(x, 4)
    assert s.shape == (4, 3) and s[1, 2] == 2 * 4 + 1
    assert returns.split_batches(x, 4, drop_last=True).shape == (3, 3)


# ----------------------------------------------------------------------------------------------
# the TensorFlow-side formulas: oracle restatement vs the reference's own sources EXECUTED under the
# torch-float64 tf stand-in (oracle/gen_golden_tf.py -> tests/golden/tf_*.npz)
# ----------------------------------------------------------------------------------------------
def _close(got, ref, tol=1e-11):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()


def test_ppo_categorical_loss_matches_executed_reference(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "tf_ppo_cat_*.npz")))
    assert len(files) == 4
    for f in files:
        g = np.load(f)
        loss, dlg, dv, parts = nets.ppo_loss_and_grads(
            g["logits"].astype(np.float64), g["value"].astype(np.float64), g["action"], g["old_logp"],
            g["adv"].astype(np.float64), g["old_v"].astype(np.float64), g["target_v"].astype(np.float64),
            float(g["clip"]), float(g["ent_coef"]), float(g["vf_clip"]), float(g["critic_coef"]))
        _close(loss, g["loss"]); _close(parts["actor_loss"], g["actor_loss"]); _close(parts["critic_loss"], g["critic_loss"])
        _close(parts["logp"], g["logp"]); _close(dlg, g["dlogits"]); _close(dv, g["dvalue"])
        p, logp_all, ent = nets.softmax_stats(g["logits"].astype(np.float64))
        _close(ent, g["entropy"])
        # the fixture really contains the tie rows it claims: ratio == 1 and |v - old_v| == VF_CLIP exactly
        assert np.array_equal(parts["logp"][:4], g["old_logp"][:4])
        assert (np.abs(g["value"][4:8].astype(np.float64) - g["old_v"][4:8]) == float(g["vf_clip"])).all()


def test_keras_impala_loss_matches_executed_reference(golden_dir):
    """oracle.nets.keras_impala_loss_and_grads against the reference's two Keras-form ``impala_loss`` closures
    (impala_cnn.py:99-108: K.mean(..., 1) per sample; impala_mlp.py:84-93: K.mean over everything) executed
    verbatim under a torch-float64 Keras-backend stand-in: both reduce to the same mean over batch and actions."""
    files = sorted(glob.glob(os.path.join(golden_dir, "tf_keras_impala_*.npz")))
    assert len(files) == 6
    for f in files:
        g = np.load(f)
        loss, dl, dv, (l_pi, l_v) = nets.keras_impala_loss_and_grads(
            g["logits"].astype(np.float64), g["value"].astype(np.float64), g["adv"].astype(np.float64),
            g["onehot"].astype(np.float64), g["target_v"].astype(np.float64), float(g["ent_coef"]))
        _close(loss, g["loss"]); _close(l_pi, g["loss_pi"]); _close(l_v, g["loss_v"])
        _close(dl, g["dlogits"]); _close(dv, g["dvalue"])
        assert g["per_sample"].shape == ((g["logits"].shape[0],) if "_cnn_" in f else (1,))


def test_ppo_gaussian_loss_matches_executed_reference(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "tf_ppo_gauss_*.npz")))
    assert len(files) == 3
    for f in files:
        g = np.load(f)
        loss, dm, dv, dls, parts = nets.gauss_ppo_loss_and_grads(
            g["mean"].astype(np.float64), g["log_std"].astype(np.float64), g["value"].astype(np.float64),
            g["action"].astype(np.float64), g["old_logp"], g["adv"].astype(np.float64), g["old_v"].astype(np.float64),
            g["target_v"].astype(np.float64), float(g["clip"]), float(g["ent_coef"]), float(g["vf_clip"]),
            float(g["critic_coef"]))
        _close(loss, g["loss"]); _close(parts["logp"], g["logp"]); _close(dm, g["dmean"]); _close(dv, g["dvalue"])
        _close(dls, g["dlog_std"]); _close(parts["actor_loss"], g["actor_loss"])


def test_impala_vtrace_and_loss_match_executed_reference(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "tf_impala_*.npz")))
    assert len(files) == 8
    for f in files:
        g = np.load(f)
        loss, dlg, dbl, parts = nets.impala_loss_and_grads(
            g["logits"], g["baseline"], g["bp_logits"], g["actions"], g["dones"], g["rewards"], int(g["batch_step"]),
            gamma=float(g["gamma"]))
        _close(parts["vs"], g["vs"]); _close(parts["pg_adv"], g["pg_adv"])
        _close(loss, g["loss"], 1e-10); _close(dlg, g["dlogits"]); _close(dbl, g["dbaseline"])
        # split_batches index behaviour: the bootstrap row of every trajectory gets exactly zero gradient
        t = int(g["batch_step"])
        assert (g["dlogits"].reshape(-1, t, g["dlogits"].shape[-1])[:, -1] == 0).all()
        assert (g["dbaseline"].reshape(-1, t)[:, -1] == 0).all()
