"""GPU parity tests (run with -m gpu on an MI355X): every C-ABI kernel entry point against the
CPU oracle on seeded inputs.  Tolerances: bit-exact for GAE (float64, index/mask ops);
fp32 kernels vs the float64 oracle <= 1e-4 relative (north_star), in practice ~1e-6."""
import ctypes
import glob
import os

import numpy as np
import pytest
import torch

from oracle import nets, returns

pytestmark = pytest.mark.gpu

RTOL = 1e-4


@pytest.fixture(scope="module")
def L():
    from xingtian_amd import lib
    lib.require_gpu()
    lib.load()
    return lib


_KEEP = []


@pytest.fixture(autouse=True)
def _keepalive():
    """device temporaries passed as raw pointers must outlive the (asynchronous) kernels"""
    _KEEP.clear()
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    t = t.cuda()
    _KEEP.append(t)
    return t


def rel_err(got, ref):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    denom = np.linalg.norm(ref.ravel()) + 1e-30
    return np.linalg.norm((got - ref).ravel()) / denom


def max_err_scaled(got, ref):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)


def geom_of(L, lay):
    g = L.ConvGeom()
    if lay.kind == "conv":
        g.H, g.W, g.C, g.KH, g.KW, g.S = lay.in_h, lay.in_w, lay.cin, lay.k, lay.k, lay.s
        g.PT, g.PL, g.OH, g.OW = lay.pt, lay.pl, lay.out_h, lay.out_w
    else:
        g.H = g.W = g.KH = g.KW = g.S = 1
        g.C = lay.cin
        g.PT = g.PL = 0
        g.OH = g.OW = 1
    g.N = lay.cout
    g.act = L.ACT[lay.act]
    return g


# ------------------------------------------------------------------ GAE
def test_gae_goldens_bit_exact(L, golden_dir):
    from xingtian_amd import ops
    files = sorted(glob.glob(os.path.join(golden_dir, "gae_*.npz")))
    assert files
    for f in files:
        g = np.load(f)
        adv, ov, tgt = ops.gae(g["value"].reshape(1, -1), g["reward"].reshape(1, -1), g["done"].reshape(1, -1),
                               float(g["gamma"]), float(g["lam"]))
        assert np.array_equal(adv.reshape(-1, 1), g["adv"]), f
        assert np.array_equal(tgt.reshape(-1, 1), g["target_value"]), f
        assert np.array_equal(ov.reshape(-1, 1), g["old_value"]), f


@pytest.mark.parametrize("n,t", [(300, 128), (5, 1024), (3, 1025), (4, 700)])
def test_gae_many_trajectories_bit_exact(L, n, t):
    """T <= 1024: one workgroup per trajectory (parallel loads, serial recurrence from LDS); longer trajectories: the
    one-thread-per-trajectory kernel; T = 700 walks the 16-step register chunks with a ragged tail."""
    from xingtian_amd import ops
    rng = np.random.default_rng(11)
    value = rng.standard_normal((n, t + 1)).astype(np.float32)
    reward = rng.choice([-1.0, 0.0, 1.0], size=(n, t), p=[0.05, 0.9, 0.05])
    reward[::7] = rng.standard_normal((len(reward[::7]), t))   # unclipped rewards too
    done = rng.random((n, t)) < 0.02
    adv, ov, tgt = ops.gae(value, reward, done)
    for i in range(n):
        a, o, tg = returns.gae(value[i].reshape(-1, 1), reward[i].copy(), done[i])
        assert np.array_equal(adv[i], a[:, 0])
        assert np.array_equal(tgt[i], tg[:, 0])
        assert np.array_equal(ov[i], o[:, 0])


# ------------------------------------------------------------------ single layers
LAYER_CASES = [
    # name, kind, in_hw, cin, cout, k, s, padding, act, B, u8
    ("ppo_conv1_u8", "conv", (84, 84), 4, 32, 8, 4, "valid", "relu", 5, True),
    ("ppo_conv2", "conv", (20, 20), 32, 32, 4, 2, "valid", "relu", 7, False),
    ("ppo_conv3", "conv", (9, 9), 32, 64, 3, 1, "valid", "relu", 9, False),
    ("ppo_fc", "dense", (1, 1), 3136, 256, 1, 1, "valid", "relu", 37, False),
    ("imp_conv1_same_u8", "conv", (84, 84), 4, 16, 8, 4, "same", "relu", 3, True),
    ("imp_conv2_same", "conv", (21, 21), 16, 32, 4, 2, "same", "relu", 6, False),
    ("imp_conv3_11", "conv", (11, 11), 32, 256, 11, 1, "valid", "relu", 40, False),
    ("imp42_conv1_same_u8", "conv", (42, 42), 4, 16, 4, 2, "same", "relu", 4, True),
    ("mlp_in4_tanh", "dense", (1, 1), 4, 64, 1, 1, "valid", "tanh", 200, False),
    ("mlp_64_tanh", "dense", (1, 1), 64, 64, 1, 1, "valid", "tanh", 130, False),
    ("conv5_s1", "conv", (15, 15), 4, 32, 5, 1, "valid", "none", 3, False),
    ("conv3_s3_same", "conv", (10, 10), 8, 12, 3, 3, "same", "tanh", 3, False),
    # the other monotonic entries of ACTIVATION_MAP (xt/model/model_utils.py:8-20), one per kernel family
    ("fc_sigmoid", "dense", (1, 1), 64, 64, 1, 1, "valid", "sigmoid", 130, False),
    ("fc_softsign", "dense", (1, 1), 3136, 256, 1, 1, "valid", "softsign", 37, False),
    ("conv2_softplus", "conv", (20, 20), 32, 32, 4, 2, "valid", "softplus", 7, False),
    ("conv3_leaky", "conv", (9, 9), 32, 64, 3, 1, "valid", "leaky_relu", 9, False),
    ("conv1_u8_elu", "conv", (84, 84), 4, 32, 8, 4, "valid", "elu", 5, True),
    ("imp_conv2_same_selu", "conv", (21, 21), 16, 32, 4, 2, "same", "selu", 6, False),
    # not monotonic: the input-gradient epilogue is given the producer's PRE-activation
    ("conv3_swish", "conv", (9, 9), 32, 64, 3, 1, "valid", "swish", 9, False),
    ("fc_gelu", "dense", (1, 1), 64, 64, 1, 1, "valid", "gelu", 130, False),
]


def _layer_data(case, seed=0):
    name, kind, in_hw, cin, cout, k, s, padding, act, b, u8 = case
    rng = np.random.default_rng(seed)
    lay = nets.LayerSpec(name, kind, cin, cout, act if act != "none" else None, k, s, padding, in_hw)
    shape = (b, in_hw[0], in_hw[1], cin)
    if u8:
        x_raw = rng.integers(0, 256, shape).astype(np.uint8)
        x = x_raw.astype(np.float64) / 255.0
    else:
        x_raw = rng.standard_normal(shape).astype(np.float32)
        x = x_raw.astype(np.float64)
    kk = k * k * cin if kind == "conv" else cin
    w = (rng.standard_normal((kk, cout)) / np.sqrt(kk)).astype(np.float32)
    bias = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    return lay, x_raw, x, w, bias, rng


@pytest.mark.parametrize("case", LAYER_CASES, ids=[c[0] for c in LAYER_CASES])
@pytest.mark.parametrize("ksplit", [1, 3])
def test_layer_fwd(L, case, ksplit):
    lay, x_raw, x, w, bias, rng = _layer_data(case)
    b = x.shape[0]
    cols = nets.im2col(x, lay) if lay.kind == "conv" else x.reshape(b, -1)
    ref = nets.act_fwd(cols @ w.astype(np.float64) + bias, lay.act)
    g = geom_of(L, lay)
    u8 = x_raw.dtype == np.uint8
    xf = L.InputXform(1 if u8 else 0, 0.0, 255.0 if u8 else 1.0)
    lib = L.load()
    m = b * lay.out_h * lay.out_w
    y = torch.full((m, lay.cout), float("nan"), device="cuda")
    partial = torch.zeros((8, m, lay.cout), device="cuda")
    L.check(lib.xt_layer_fwd(ctypes.byref(g), ctypes.byref(xf), b, L.ptr(dev(x_raw)), None, L.ptr(dev(w)),
                             L.ptr(dev(bias)), L.ptr(y), L.ptr(partial), ksplit, None), "fwd")
    torch.cuda.synchronize()
    got = y.cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_err(got, ref) < 2e-6, case[0]
    assert max_err_scaled(got, ref) < 1e-5


def test_layer_fwd_with_row_gather_and_mean_std(L):
    case = ("imp42", "conv", (42, 42), 4, 16, 4, 2, "same", "relu", 6, True)
    lay, x_raw, _, w, bias, rng = _layer_data(case, seed=3)
    pool = rng.integers(0, 256, (20, 42, 42, 4)).astype(np.uint8)
    idx = np.array([7, 7, 0, 19, 3, 12], np.int32)
    x = (pool[idx].astype(np.float64) - 128.0) / 128.0
    ref = nets.act_fwd(nets.im2col(x, lay) @ w.astype(np.float64) + bias, "relu")
    g = geom_of(L, lay)
    xf = L.InputXform(1, 128.0, 128.0)
    y = torch.zeros((6 * 21 * 21, 16), device="cuda")
    L.check(L.load().xt_layer_fwd(ctypes.byref(g), ctypes.byref(xf), 6, L.ptr(dev(pool)), L.ptr(dev(idx)),
                                  L.ptr(dev(w)), L.ptr(dev(bias)), L.ptr(y), None, 1, None), "fwd")
    assert rel_err(y.cpu().numpy(), ref) < 2e-6


@pytest.mark.parametrize("case", LAYER_CASES, ids=[c[0] for c in LAYER_CASES])
@pytest.mark.parametrize("msplit", [1, 5])
def test_layer_wgrad(L, case, msplit):
    lay, x_raw, x, w, bias, rng = _layer_data(case, seed=1)
    b = x.shape[0]
    cols = nets.im2col(x, lay) if lay.kind == "conv" else x.reshape(b, -1)
    m = cols.shape[0]
    dy = rng.standard_normal((m, lay.cout)).astype(np.float32)
    ref_w = cols.T @ dy.astype(np.float64)
    ref_b = dy.astype(np.float64).sum(0)
    g = geom_of(L, lay)
    u8 = x_raw.dtype == np.uint8
    xf = L.InputXform(1 if u8 else 0, 0.0, 255.0 if u8 else 1.0)
    kk = cols.shape[1]
    out = torch.full(((kk + 1) * lay.cout,), float("nan"), device="cuda")
    slabs = torch.zeros((16 * (kk + 1) * lay.cout,), device="cuda")
    L.check(L.load().xt_layer_wgrad(ctypes.byref(g), ctypes.byref(xf), b, L.ptr(dev(x_raw)), None, L.ptr(dev(dy)),
                                    L.ptr(out), L.ptr(slabs), msplit, None), "wgrad")
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_err(got[:kk * lay.cout].reshape(kk, lay.cout), ref_w) < 3e-6, case[0]
    assert rel_err(got[kk * lay.cout:], ref_b) < 3e-6


@pytest.mark.parametrize("case", LAYER_CASES, ids=[c[0] for c in LAYER_CASES])
def test_layer_dgrad(L, case):
    lay, x_raw, x, w, bias, rng = _layer_data(case, seed=2)
    if x_raw.dtype == np.uint8:
        x_raw = rng.standard_normal(x_raw.shape).astype(np.float32)   # dgrad needs an fp32 producer output
    b = x_raw.shape[0]
    m = b * lay.out_h * lay.out_w
    dy = rng.standard_normal((m, lay.cout)).astype(np.float32)
    dcols = dy.astype(np.float64) @ w.astype(np.float64).T
    dx = nets.col2im(dcols, lay, b) if lay.kind == "conv" else dcols.reshape(x_raw.shape)
    extra = (lay.act,) if lay.act not in (None, "relu", "tanh") else ()     # the producer's activation derivative
    for act_prev in ("relu", "tanh") + extra:
        xp = x_raw if act_prev == "relu" else nets.act_fwd(x_raw.astype(np.float64), act_prev).astype(np.float32)
        if act_prev in nets.NEEDS_PREACT:       # the kernel is handed the producer's pre-activation for these
            xp = x_raw
            ref = nets.act_bwd(dx.reshape(xp.shape), None, act_prev, xp.astype(np.float64))
        else:
            ref = nets.act_bwd(dx.reshape(xp.shape), xp.astype(np.float64), act_prev)
        g = geom_of(L, lay)
        out = torch.full(xp.shape, float("nan"), device="cuda")
        L.check(L.load().xt_layer_dgrad(ctypes.byref(g), b, L.ptr(dev(dy)), L.ptr(dev(w)), L.ptr(dev(xp)),
                                        L.ACT[act_prev], L.ptr(out), None), "dgrad")
        got = out.cpu().numpy()
        assert np.isfinite(got).all(), case[0]
        # (softplus / softsign: the fp32 output carries the pre-activation only to ~1e-7 relative near saturation)
        assert rel_err(got, ref) < (3e-6 if not extra or act_prev in ("relu", "tanh") else 2e-5), (case[0], act_prev)


# ------------------------------------------------------------------ heads / losses
@pytest.mark.parametrize("a_dim,shared", [(4, True), (6, True), (2, False), (18, True)])
def test_heads_and_ppo_loss(L, a_dim, shared):
    rng = np.random.default_rng(5)
    b, f = 77, 256 if shared else 64
    f_pi = rng.standard_normal((b, f)).astype(np.float32)
    f_pi[f_pi < -0.5] = 0.0
    f_v = f_pi if shared else np.tanh(rng.standard_normal((b, f))).astype(np.float32)
    wpi = (rng.standard_normal((f, a_dim)) * 0.1).astype(np.float32)
    bpi = (rng.standard_normal(a_dim) * 0.1).astype(np.float32)
    wv = (rng.standard_normal((f, 1)) * 0.1).astype(np.float32)
    bv = np.array([0.3], np.float32)
    lib = L.load()
    d_fpi, d_fv = dev(f_pi), None
    d_fv = d_fpi if shared else dev(f_v)
    logits = torch.zeros((b, a_dim), device="cuda")
    value = torch.zeros((b,), device="cuda")
    L.check(lib.xt_heads_fwd(L.ptr(d_fpi), L.ptr(d_fv), b, f, a_dim, L.ptr(dev(wpi)), L.ptr(dev(bpi)), L.ptr(dev(wv)),
                             L.ptr(dev(bv)), L.ptr(logits), L.ptr(value), None), "heads_fwd")
    ref_logits = f_pi.astype(np.float64) @ wpi + bpi
    ref_value = f_v.astype(np.float64) @ wv + bv
    assert rel_err(logits.cpu().numpy(), ref_logits) < 2e-6
    assert rel_err(value.cpu().numpy(), ref_value[:, 0]) < 2e-6

    # PPO loss on the oracle's logits (isolates the loss kernel)
    n_pool = 150
    idx = rng.permutation(n_pool)[:b].astype(np.int32)
    action = rng.integers(0, a_dim, n_pool).astype(np.int32)
    old_logp = (-np.abs(rng.standard_normal(n_pool)) - 0.3).astype(np.float32)
    adv = rng.standard_normal(n_pool)
    old_v = rng.standard_normal(n_pool).astype(np.float32)
    target_v = old_v + rng.standard_normal(n_pool) * 4
    lg32, v32 = ref_logits.astype(np.float32), ref_value.astype(np.float32)
    clip, entc, vfc, cc = 0.1, 0.003, 0.5, 0.7
    loss, dlg, dv, parts = nets.ppo_loss_and_grads(
        lg32.astype(np.float64), v32.astype(np.float64), action[idx], old_logp[idx].reshape(-1, 1).astype(np.float64),
        adv[idx].astype(np.float32).reshape(-1, 1).astype(np.float64), old_v[idx].reshape(-1, 1).astype(np.float64),
        target_v[idx].astype(np.float32).reshape(-1, 1).astype(np.float64), clip, entc, vfc, cc)
    dlogits = torch.zeros((b, a_dim), device="cuda")
    dvalue = torch.zeros((b,), device="cuda")
    terms = torch.zeros((b, 4), device="cuda")
    out = torch.zeros(8, device="cuda")
    acc = torch.zeros(8, device="cuda")
    L.check(lib.xt_ppo_loss(L.ptr(dev(lg32)), L.ptr(dev(v32[:, 0])), b, a_dim, L.ptr(dev(idx)), L.ptr(dev(action)),
                            L.ptr(dev(old_logp)), L.ptr(dev(adv)), L.ptr(dev(old_v)), L.ptr(dev(target_v)),
                            clip, entc, vfc, cc, 1.0 / b, L.ptr(dlogits), L.ptr(dvalue), L.ptr(terms), None), "ppo_loss")
    L.check(lib.xt_ppo_loss_reduce(L.ptr(terms), b, entc, cc, 1.0 / b, L.ptr(out), L.ptr(acc), None), "reduce")
    o = out.cpu().numpy()
    assert abs(o[0] - loss) < 1e-5 * max(1.0, abs(loss))
    assert abs(o[1] - parts["actor_loss"]) < 1e-5 and abs(o[2] - parts["critic_loss"]) < 1e-5 * max(1, parts["critic_loss"])
    assert rel_err(dlogits.cpu().numpy(), dlg) < 1e-5
    assert rel_err(dvalue.cpu().numpy(), dv[:, 0]) < 1e-5
    assert acc.cpu().numpy()[1] == 1.0

    # heads backward
    for act_prev in ("relu", "tanh"):
        dl32, dv32 = dlg.astype(np.float32), dv.astype(np.float32)
        dwpi = torch.zeros((f, a_dim), device="cuda"); dbpi = torch.zeros(a_dim, device="cuda")
        dwv = torch.zeros((f,), device="cuda"); dbv = torch.zeros(1, device="cuda")
        df_pi = torch.zeros((b, f), device="cuda")
        df_v = df_pi if shared else torch.zeros((b, f), device="cuda")
        L.check(lib.xt_heads_bwd(L.ptr(d_fpi), L.ptr(d_fv), b, f, a_dim, L.ptr(dev(wpi)), L.ptr(dev(wv)),
                                 L.ptr(dev(dl32)), L.ptr(dev(dv32[:, 0])), L.ACT[act_prev], L.ptr(dwpi), L.ptr(dbpi),
                                 L.ptr(dwv), L.ptr(dbv), L.ptr(df_pi), L.ptr(df_v), None), "heads_bwd")
        assert rel_err(dwpi.cpu().numpy(), f_pi.astype(np.float64).T @ dl32) < 3e-6
        assert rel_err(dbpi.cpu().numpy(), dl32.astype(np.float64).sum(0)) < 3e-6
        assert rel_err(dwv.cpu().numpy(), (f_v.astype(np.float64).T @ dv32)[:, 0]) < 3e-6
        assert rel_err(dbv.cpu().numpy(), dv32.astype(np.float64).sum(0)) < 3e-6
        dpi = dl32.astype(np.float64) @ wpi.T
        dvf = dv32.astype(np.float64) @ wv.T
        if shared:
            ref = nets.act_bwd(dpi + dvf, f_pi.astype(np.float64), act_prev)
            assert rel_err(df_pi.cpu().numpy(), ref) < 3e-6
        else:
            assert rel_err(df_pi.cpu().numpy(), nets.act_bwd(dpi, f_pi.astype(np.float64), act_prev)) < 3e-6
            assert rel_err(df_v.cpu().numpy(), nets.act_bwd(dvf, f_v.astype(np.float64), act_prev)) < 3e-6


@pytest.mark.parametrize("tlen,n_traj,a_dim", [(128, 3, 4), (50, 20, 6), (2, 1, 4), (5, 2, 18)])
def test_impala_vtrace_loss(L, tlen, n_traj, a_dim):
    rng = np.random.default_rng(9)
    n = tlen * n_traj
    logits = rng.standard_normal((n, a_dim)).astype(np.float32)
    baseline = rng.standard_normal(n).astype(np.float32)
    bp = rng.standard_normal((n, a_dim)).astype(np.float32)
    act = rng.integers(0, a_dim, n).astype(np.int32)
    done = rng.random(n) < 0.1
    rew = (rng.standard_normal(n) * 2).astype(np.float32)
    loss, dlg, dbl, parts = nets.impala_loss_and_grads(logits, baseline, bp, act, done, rew, tlen)
    dlogits = torch.full((n, a_dim), float("nan"), device="cuda")
    dbase = torch.full((n,), float("nan"), device="cuda")
    out = torch.zeros(8 + n_traj, device="cuda")
    vs = torch.zeros((n_traj, tlen - 1), device="cuda")
    pg = torch.zeros((n_traj, tlen - 1), device="cuda")
    L.check(L.load().xt_impala_loss(L.ptr(dev(logits)), L.ptr(dev(baseline)), L.ptr(dev(bp)), L.ptr(dev(act)),
                                    L.ptr(dev(done.astype(np.uint8))), L.ptr(dev(rew)), n_traj, tlen, a_dim, 0.99,
                                    L.ptr(dlogits), L.ptr(dbase), L.ptr(out), None, L.ptr(vs), L.ptr(pg), None),
            "impala_loss")
    assert rel_err(vs.cpu().numpy(), parts["vs"].T) < 1e-5      # oracle is [T-1, B]
    assert rel_err(pg.cpu().numpy(), parts["pg_adv"].T) < 1e-5
    assert abs(out[0].item() - loss) < 2e-5 * max(1.0, abs(loss))
    assert rel_err(dlogits.cpu().numpy(), dlg) < 1e-5
    assert rel_err(dbase.cpu().numpy(), dbl) < 1e-5
    # bit-exact index/mask behaviour: the bootstrap step of every trajectory carries no gradient
    assert (dlogits.cpu().numpy().reshape(n_traj, tlen, a_dim)[:, -1] == 0).all()
    assert (dbase.cpu().numpy().reshape(n_traj, tlen)[:, -1] == 0).all()


# ------------------------------------------------------------------ optimiser
@pytest.mark.parametrize("count,clip", [(847496, 5.0), (1003, 0.01), (4, 40.0)])
def test_adam_tf_clip(L, count, clip):
    rng = np.random.default_rng(4)
    p = rng.standard_normal(count).astype(np.float32)
    params = {"w": p.copy()}
    opt = nets.AdamTF(params, 2.5e-4)
    dp, dm, dv_ = dev(p), torch.zeros(count, device="cuda"), torch.zeros(count, device="cuda")
    state = torch.zeros(8, device="cuda")
    scratch = torch.zeros(1024, device="cuda")
    lib = L.load()
    L.check(lib.xt_adam_state_init(L.ptr(state), None), "init")
    for it in range(4):
        g = (rng.standard_normal(count) * (0.01 if it % 2 else 1.0)).astype(np.float32)
        clipped, gn = nets.clip_by_global_norm({"w": g}, clip)
        opt.apply(params, clipped)
        L.check(lib.xt_adam_tf_clip(L.ptr(dp), L.ptr(dev(g)), L.ptr(dm), L.ptr(dv_), count, 2.5e-4, 0.9, 0.999, 1e-8,
                                    clip, 1.0, L.ptr(state), L.ptr(scratch), None), "adam")
        st = state.cpu().numpy()
        assert abs(st[4] - gn) < 1e-5 * gn
        assert st[5] == it + 1
        np.testing.assert_allclose(dp.cpu().numpy(), params["w"], rtol=2e-5, atol=2e-7)


@pytest.mark.parametrize("b,a_dim", [(37, 1), (200, 3), (64, 6)])
def test_ppo_loss_gauss_vs_oracle(L, b, a_dim):
    """xt_ppo_loss_gauss (DiagGaussianDist, tf_dist.py:47-87) against the oracle on the same fp32 inputs:
    loss scalars, d mean, d value and the per-sample pi_logstd rows (whose column sums are the gradient)."""
    lib = L.load()
    rng = np.random.default_rng(100 + b)
    n_pool = b + 29
    idx = rng.permutation(n_pool)[:b].astype(np.int32)
    mean = rng.standard_normal((b, a_dim)).astype(np.float32)
    log_std = (rng.standard_normal(a_dim) * 0.4).astype(np.float32)
    value = rng.standard_normal(b).astype(np.float32)
    action = (rng.standard_normal((n_pool, a_dim)) * 1.3).astype(np.float32)
    old_logp = (-np.abs(rng.standard_normal(n_pool)) - 0.3 * a_dim).astype(np.float32)
    adv = rng.standard_normal(n_pool)
    old_v = rng.standard_normal(n_pool).astype(np.float32)
    target_v = old_v + rng.standard_normal(n_pool) * 2
    clip, entc, vfc, cc = 0.2, 0.01, 0.5, 1.0
    f64 = lambda x: np.asarray(x, np.float32).astype(np.float64)
    loss, dmean, dv, dls, parts = nets.gauss_ppo_loss_and_grads(
        f64(mean), f64(log_std).reshape(1, -1), f64(value).reshape(-1, 1), f64(action[idx]),
        f64(old_logp[idx]).reshape(-1, 1), f64(adv[idx]).reshape(-1, 1), f64(old_v[idx]).reshape(-1, 1),
        f64(target_v[idx]).reshape(-1, 1), clip, entc, vfc, cc)
    d_dmean = torch.zeros((b, a_dim), device="cuda")
    d_dv = torch.zeros(b, device="cuda")
    d_rows = torch.zeros((b, a_dim), device="cuda")
    terms = torch.zeros((b, 4), device="cuda")
    out = torch.zeros(8, device="cuda")
    L.check(lib.xt_ppo_loss_gauss(L.ptr(dev(mean)), L.ptr(dev(log_std)), L.ptr(dev(value)), b, a_dim, L.ptr(dev(idx)),
                                  L.ptr(dev(action)), L.ptr(dev(old_logp)), L.ptr(dev(adv)), L.ptr(dev(old_v)),
                                  L.ptr(dev(target_v)), clip, entc, vfc, cc, 1.0 / b, L.ptr(d_dmean), L.ptr(d_dv),
                                  L.ptr(d_rows), L.ptr(terms), None), "ppo_loss_gauss")
    L.check(lib.xt_ppo_loss_reduce(L.ptr(terms), b, entc, cc, 1.0 / b, L.ptr(out), None, None), "reduce")
    o = out.cpu().numpy()
    assert abs(o[0] - loss) < 1e-5 * max(1.0, abs(loss))
    assert abs(o[3] - parts["entropy"]) < 1e-5 * max(1.0, abs(parts["entropy"]))
    assert rel_err(d_dmean.cpu().numpy(), dmean) < 1e-5
    assert rel_err(d_dv.cpu().numpy(), dv[:, 0]) < 1e-5
    assert rel_err(d_rows.cpu().numpy(), parts["dls_rows"]) < 1e-5
    assert rel_err(d_rows.cpu().numpy().astype(np.float64).sum(0), dls[0]) < 1e-5


@pytest.mark.parametrize("name,in_hw,cin,cout,k,s", [("ppo_conv2", (20, 20), 32, 32, 4, 2), ("ppo_conv3", (9, 9), 32, 64, 3, 1)])
@pytest.mark.parametrize("b", [128, 160, 256, 320])
def test_layer_fwd_and_dgrad_at_benchmark_batches(L, name, in_hw, cin, cout, k, s, b):
    """The register-direct kernels (xt_direct.hip) are only routed to for large tile counts and pick their
    wave/reduction split from the batch: cover BASELINE.json's B = 320 and the data-parallel shard sizes of it
    through the same C-ABI entry points as the small geometry cases above."""
    case = (name, "conv", in_hw, cin, cout, k, s, "valid", "relu", b, False)
    lay, x_raw, x, w, bias, rng = _layer_data(case, seed=3)
    g = geom_of(L, lay)
    lib = L.load()
    m = b * lay.out_h * lay.out_w
    cols = nets.im2col(x.astype(np.float64).reshape(b, lay.in_h, lay.in_w, lay.cin), lay)
    ref = nets.act_fwd(cols @ w.astype(np.float64).reshape(-1, lay.cout) + bias, lay.act)
    out = torch.full((m, lay.cout), float("nan"), device="cuda")
    part = torch.zeros(16 * m * lay.cout, device="cuda")
    L.check(lib.xt_layer_fwd(ctypes.byref(g), None, b, L.ptr(dev(x_raw)), None, L.ptr(dev(w)),
                             L.ptr(dev(bias.astype(np.float32))), L.ptr(out), L.ptr(part), 1, None), "fwd")
    got = out.cpu().numpy()
    assert np.isfinite(got).all() and rel_err(got, ref) < 3e-6
    dy = rng.standard_normal((m, lay.cout)).astype(np.float32)
    dx = nets.col2im(dy.astype(np.float64) @ w.astype(np.float64).reshape(-1, lay.cout).T, lay, b)
    for act_prev in ("relu", "tanh"):
        xp = x_raw if act_prev == "relu" else np.tanh(x_raw).astype(np.float32)
        refd = nets.act_bwd(dx.reshape(xp.shape), xp.astype(np.float64), act_prev)
        outd = torch.full(xp.shape, float("nan"), device="cuda")
        L.check(lib.xt_layer_dgrad(ctypes.byref(g), b, L.ptr(dev(dy)), L.ptr(dev(w)), L.ptr(dev(xp)),
                                   L.ACT[act_prev], L.ptr(outd), None), "dgrad")
        gotd = outd.cpu().numpy()
        assert np.isfinite(gotd).all() and rel_err(gotd, refd) < 3e-6, (name, b, act_prev)


@pytest.mark.parametrize("b,a_dim", [(128, 4), (44, 18), (1, 3)])
def test_keras_impala_loss_vs_oracle(L, b, a_dim):
    """xt_keras_impala_loss (softmax + Keras impala_loss + 0.5 mse, label rows gathered through idx) against the
    float64 oracle: loss terms, d/dlogits, d/dvalue."""
    rng = np.random.default_rng(70 + b)
    n = b + 30
    logits = rng.standard_normal((b, a_dim)).astype(np.float32) * 2
    value = rng.standard_normal(b).astype(np.float32)
    adv = rng.standard_normal(n).astype(np.float32)
    onehot = np.eye(a_dim, dtype=np.float32)[rng.integers(0, a_dim, n)]
    tv = rng.standard_normal(n).astype(np.float32)
    idx = rng.permutation(n)[:b].astype(np.int32)
    loss, dl, dv, parts = nets.keras_impala_loss_and_grads(logits.astype(np.float64), value.astype(np.float64).reshape(b, 1),
                                                           adv[idx].astype(np.float64), onehot[idx].astype(np.float64),
                                                           tv[idx].astype(np.float64), 0.01)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    d_dl = torch.empty((b, a_dim), dtype=torch.float32, device="cuda")
    d_dv = torch.empty((b,), dtype=torch.float32, device="cuda")
    out = torch.zeros((4 + 2 * b,), dtype=torch.float32, device="cuda")
    acc = torch.zeros((2,), dtype=torch.float32, device="cuda")
    lib = L.load()
    dev = [d(x) for x in (logits, value, idx, adv, onehot, tv)]        # keep the uploads alive across the launch
    L.check(lib.xt_keras_impala_loss(L.ptr(dev[0]), L.ptr(dev[1]), b, a_dim, L.ptr(dev[2]), L.ptr(dev[3]),
                                     L.ptr(dev[4]), L.ptr(dev[5]), 0.01, L.ptr(d_dl), L.ptr(d_dv), L.ptr(out), L.ptr(acc),
                                     L.stream_ptr()), "xt_keras_impala_loss")
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert abs(o[0] - loss) < 1e-5 * max(1, abs(loss)) and abs(o[1] - parts[0]) < 1e-5 * max(1, abs(parts[0]))
    assert abs(o[2] - parts[1]) < 1e-5 * max(1, abs(parts[1]))
    assert np.allclose(acc.cpu().numpy(), [o[0] * b, b], rtol=1e-6)
    assert np.linalg.norm(d_dl.cpu().numpy() - dl) <= 1e-5 * np.linalg.norm(dl)
    assert np.linalg.norm(d_dv.cpu().numpy() - dv.reshape(-1)) <= 1e-5 * np.linalg.norm(dv)


def test_keras_impala_loss_kernel_vs_executed_reference(L, golden_dir):
    """xt_keras_impala_loss against the reference's own Keras-form closures executed under the torch-float64
    backend stand-in (tests/golden/tf_keras_impala_*.npz; impala_cnn.py:99-108, impala_mlp.py:84-93), including the
    peaked-policy cases where the 1e-10 epsilon decides the value.  fp32 kernel vs float64 golden."""
    files = sorted(glob.glob(os.path.join(golden_dir, "tf_keras_impala_*.npz")))
    assert len(files) == 6
    lib = L.load()
    for f in files:
        g = np.load(f)
        b, a_dim = g["logits"].shape
        d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
        dev = [d(g["logits"]), d(g["value"].reshape(-1)), d(np.arange(b, dtype=np.int32)), d(g["adv"].reshape(-1)),
               d(g["onehot"]), d(g["target_v"].reshape(-1))]
        d_dl = torch.empty((b, a_dim), dtype=torch.float32, device="cuda")
        d_dv = torch.empty((b,), dtype=torch.float32, device="cuda")
        out = torch.zeros((4 + 2 * b,), dtype=torch.float32, device="cuda")
        acc = torch.zeros((2,), dtype=torch.float32, device="cuda")
        L.check(lib.xt_keras_impala_loss(L.ptr(dev[0]), L.ptr(dev[1]), b, a_dim, L.ptr(dev[2]), L.ptr(dev[3]),
                                         L.ptr(dev[4]), L.ptr(dev[5]), float(g["ent_coef"]), L.ptr(d_dl), L.ptr(d_dv),
                                         L.ptr(out), L.ptr(acc), L.stream_ptr()), "xt_keras_impala_loss")
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        for got, ref in ((o[0], g["loss"]), (o[1], g["loss_pi"]), (o[2], g["loss_v"])):
            assert abs(got - float(ref)) < 2e-5 * max(1.0, abs(float(ref))), (f, got, float(ref))
        assert np.linalg.norm(d_dl.cpu().numpy() - g["dlogits"]) <= 2e-5 * np.linalg.norm(g["dlogits"]), f
        assert np.linalg.norm(d_dv.cpu().numpy() - g["dvalue"].reshape(-1)) <= 1e-5 * np.linalg.norm(g["dvalue"]), f


# ----------------------------------------------------------------------------------------------
# the loss kernels against the reference's OWN formula sources executed under the torch-float64 tf stand-in
# (oracle/gen_golden_tf.py -> tests/golden/tf_*.npz): no restatement between the kernel and the reference
# ----------------------------------------------------------------------------------------------
def test_ppo_loss_kernel_vs_executed_reference(L, golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "tf_ppo_cat_*.npz")))
    assert len(files) == 4
    lib = L.load()
    for f in files:
        g = np.load(f)
        b, a_dim = g["logits"].shape
        clip, entc, vfc, cc = (float(g[k]) for k in ("clip", "ent_coef", "vf_clip", "critic_coef"))
        dlogits = torch.zeros((b, a_dim), device="cuda")
        dvalue = torch.zeros((b,), device="cuda")
        terms = torch.zeros((b, 4), device="cuda")
        out = torch.zeros(8, device="cuda")
        L.check(lib.xt_ppo_loss(L.ptr(dev(g["logits"])), L.ptr(dev(g["value"][:, 0])), b, a_dim, None,
                                L.ptr(dev(g["action"])), L.ptr(dev(g["old_logp"][:, 0].astype(np.float32))),
                                L.ptr(dev(g["adv"][:, 0].astype(np.float64))), L.ptr(dev(g["old_v"][:, 0])),
                                L.ptr(dev(g["target_v"][:, 0].astype(np.float64))), clip, entc, vfc, cc, 1.0 / b,
                                L.ptr(dlogits), L.ptr(dvalue), L.ptr(terms), None), "ppo_loss")
        L.check(lib.xt_ppo_loss_reduce(L.ptr(terms), b, entc, cc, 1.0 / b, L.ptr(out), None, None), "reduce")
        o = out.cpu().numpy()
        assert abs(o[0] - g["loss"]) < 1e-5 * max(1.0, abs(g["loss"])), f
        assert abs(o[1] - g["actor_loss"]) < 1e-5 * max(1.0, abs(g["actor_loss"])), f
        assert abs(o[2] - g["critic_loss"]) < 1e-5 * max(1.0, abs(g["critic_loss"])), f
        assert abs(o[3] - g["entropy"].mean()) < 1e-5, f
        assert rel_err(dlogits.cpu().numpy(), g["dlogits"]) < 1e-5, f
        assert rel_err(dvalue.cpu().numpy(), g["dvalue"][:, 0]) < 1e-5, f


def test_ppo_gauss_loss_kernel_vs_executed_reference(L, golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "tf_ppo_gauss_*.npz")))
    assert len(files) == 3
    lib = L.load()
    for f in files:
        g = np.load(f)
        b, a_dim = g["mean"].shape
        clip, entc, vfc, cc = (float(g[k]) for k in ("clip", "ent_coef", "vf_clip", "critic_coef"))
        d_dmean = torch.zeros((b, a_dim), device="cuda")
        d_dv = torch.zeros(b, device="cuda")
        d_rows = torch.zeros((b, a_dim), device="cuda")
        terms = torch.zeros((b, 4), device="cuda")
        out = torch.zeros(8, device="cuda")
        L.check(lib.xt_ppo_loss_gauss(L.ptr(dev(g["mean"])), L.ptr(dev(g["log_std"][0])), L.ptr(dev(g["value"][:, 0])), b,
                                      a_dim, None, L.ptr(dev(g["action"])),
                                      L.ptr(dev(g["old_logp"][:, 0].astype(np.float32))),
                                      L.ptr(dev(g["adv"][:, 0].astype(np.float64))), L.ptr(dev(g["old_v"][:, 0])),
                                      L.ptr(dev(g["target_v"][:, 0].astype(np.float64))), clip, entc, vfc, cc, 1.0 / b,
                                      L.ptr(d_dmean), L.ptr(d_dv), L.ptr(d_rows), L.ptr(terms), None), "ppo_loss_gauss")
        L.check(lib.xt_ppo_loss_reduce(L.ptr(terms), b, entc, cc, 1.0 / b, L.ptr(out), None, None), "reduce")
        o = out.cpu().numpy()
        assert abs(o[0] - g["loss"]) < 1e-5 * max(1.0, abs(g["loss"])), f
        assert rel_err(d_dmean.cpu().numpy(), g["dmean"]) < 1e-5, f
        assert rel_err(d_dv.cpu().numpy(), g["dvalue"][:, 0]) < 1e-5, f
        assert rel_err(d_rows.cpu().numpy().astype(np.float64).sum(0), g["dlog_std"][0]) < 1e-5, f


def test_impala_loss_kernel_vs_executed_reference(L, golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "tf_impala_*.npz")))
    assert len(files) == 8
    lib = L.load()
    for f in files:
        g = np.load(f)
        n, a_dim = g["logits"].shape
        tlen = int(g["batch_step"])
        n_traj = n // tlen
        dlogits = torch.full((n, a_dim), float("nan"), device="cuda")
        dbase = torch.full((n,), float("nan"), device="cuda")
        out = torch.zeros(8 + n_traj, device="cuda")
        vs = torch.zeros((n_traj, tlen - 1), device="cuda")
        pg = torch.zeros((n_traj, tlen - 1), device="cuda")
        L.check(lib.xt_impala_loss(L.ptr(dev(g["logits"])), L.ptr(dev(g["baseline"])), L.ptr(dev(g["bp_logits"])),
                                   L.ptr(dev(g["actions"])), L.ptr(dev(g["dones"].astype(np.uint8))),
                                   L.ptr(dev(g["rewards"])), n_traj, tlen, a_dim, float(g["gamma"]), L.ptr(dlogits),
                                   L.ptr(dbase), L.ptr(out), None, L.ptr(vs), L.ptr(pg), None), "impala_loss")
        assert rel_err(vs.cpu().numpy(), g["vs"].T) < 1e-5, f          # the reference's views are [T-1, B]
        assert rel_err(pg.cpu().numpy(), g["pg_adv"].T) < 1e-5, f
        assert abs(out[0].item() - g["loss"]) < 2e-5 * max(1.0, abs(g["loss"])), f
        assert rel_err(dlogits.cpu().numpy(), g["dlogits"]) < 1e-5, f
        assert rel_err(dbase.cpu().numpy(), g["dbaseline"]) < 1e-5, f
        # index/mask behaviour bit-exact: zero gradient exactly where the executed reference has zero gradient
        assert np.array_equal(dlogits.cpu().numpy() == 0, g["dlogits"] == 0), f
        assert (dbase.cpu().numpy().reshape(n_traj, tlen)[:, -1] == 0).all(), f


@pytest.mark.parametrize("n", [1, 2, 63, 64, 1000, 4096, 4097, 40000])
def test_adv_normalize_f64_matches_numpy(L, n):
    """ADV_NORM (the reference's commented-out line, xt/algorithm/ppo/ppo.py:73): (adv - adv.mean()) / (adv.std() + 1e-8)
    in float64 on the device with wave-level reductions; numpy sums pairwise, the kernel in a fixed strided order, so the
    bar is 1e-12 relative to the normalised scale (float64 eps x a small multiple), not bit equality."""
    rng = np.random.default_rng(n)
    adv = rng.standard_normal(n) * 3.0 + 0.7
    want = (adv - adv.mean()) / (adv.std() + 1e-8)
    d = torch.from_numpy(adv.copy()).cuda()
    stats = torch.zeros(2, dtype=torch.float64, device="cuda")
    L.check(L.load().xt_adv_normalize_f64(L.ptr(d), n, 1e-8, L.ptr(stats), None), "xt_adv_normalize_f64")
    got, st = d.cpu().numpy(), stats.cpu().numpy()
    assert abs(st[0] - adv.mean()) < 1e-13 * max(1.0, abs(adv.mean())) and abs(st[1] - adv.std()) < 1e-13 * max(1.0, adv.std())
    assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, np.max(np.abs(want)))
    d2 = torch.from_numpy(adv.copy()).cuda()              # reproducible: same bits run to run
    L.check(L.load().xt_adv_normalize_f64(L.ptr(d2), n, 1e-8, None, None), "xt_adv_normalize_f64")
    assert torch.equal(d, d2)


@pytest.mark.parametrize("b", [1, 3, 128, 256])
def test_conv1_conv2_of_a_frame_stack_in_one_launch_equals_the_two_launches(b):
    """xt_tuning.fwd_fuse12 (round 6): ImpalaCnnOpt 84x84's first two layers (uint8 8x8/4 SAME 4 -> 16, 4x4/2 SAME 16 -> 32,
    xt/model/impala/impala_cnn_opt.py:118-125) as ONE launch per frame stack -- conv1's tiles are computed exactly as by the
    separate kernel (bit for bit), conv2 comes out of LDS on fp32 MFMA with the K halves combined in fixed order (vs the
    register-direct kernel: rounding of a different summation order only); against the float64 oracle both stay <= 3e-6."""
    from xingtian_amd import lib as L
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    spec = netspec.impala_cnn_opt((84, 84, 4), 4, 0.0, 255.0, "uint8")
    rng = np.random.default_rng(40 + b)
    obs = rng.integers(0, 256, (b, 84, 84, 4)).astype(np.uint8)
    outs = []
    for knob in (0, 256):
        old = L.set_tuning(fwd_fuse12=knob)
        try:
            net = HipActorCritic(spec, max_batch=b, seed=7)
            logits, value = net.forward(obs)
            torch.cuda.synchronize()
            a1, _ = net.layer_buffers(0, b)
            a2, _ = net.layer_buffers(1, b)
            outs.append((a1.cpu().numpy().copy(), a2.cpu().numpy().copy(), logits.cpu().numpy().copy(), value.cpu().numpy().copy()))
        finally:
            L.set_tuning(**old)
    (r1, r2, rl, rv), (f1, f2, fl, fv) = outs
    assert np.abs(r2).max() > 0 and np.array_equal(r1, f1)
    assert np.linalg.norm(f2 - r2) / np.linalg.norm(r2) < 2e-6
    assert np.allclose(fl, rl, rtol=1e-5, atol=1e-6) and np.allclose(fv, rv, rtol=1e-5, atol=1e-6)
