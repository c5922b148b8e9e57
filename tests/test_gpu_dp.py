"""Multi-RANK data-parallel tests of the product path: two processes, one replica each, on
the one GPU of the test box; gradients are exchanged through a real 2-rank ``torch.distributed`` group (gloo, because
RCCL refuses two ranks on one device).  The result is compared with the single-process update on the same data:
strict sharding reproduces the single-GPU update up to fp32 summation order, the replicas stay bit-identical."""
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


def _run_two_ranks(tmp_path, mode):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path), mode]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    assert proc.returncode == 0, proc.stdout.decode()[-3000:]
    p0 = np.load(os.path.join(str(tmp_path), "params_{}_r0.npy".format(mode)))
    p1 = np.load(os.path.join(str(tmp_path), "params_{}_r1.npy".format(mode)))
    assert np.array_equal(p0, p1), "replicas diverged"
    return p0


def _delta_err(got, ref, start):
    return np.linalg.norm((got - start) - (ref - start)) / (np.linalg.norm(ref - start) + 1e-30)


def test_strict_sharding_two_ranks_equals_the_single_gpu_update(tmp_path):
    """Global minibatch of 32 rows -> 16 rows per rank (8 in the short last minibatch), shared permutations, loss
    means over the global minibatch: 6 SGD steps must reproduce ``xt_net_ppo_train`` of one process."""
    import dp_worker
    from xingtian_amd.model.hip_net import HipActorCritic
    got = _run_two_ranks(tmp_path, "strict")
    spec, cfg, n = dp_worker.ppo_case()
    obs, lab, perms = dp_worker.ppo_rollout(100, n)
    net = HipActorCritic(spec, max_batch=cfg["BATCH_SIZE"], seed=5)
    start = net.params.cpu().numpy().copy()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    net.ppo_train(net.make_ppo_cfg(cfg), net.to_device_obs(obs), d(perms), d(lab[0]), d(lab[1].reshape(-1)),
                  d(lab[2].reshape(-1)), d(lab[3].reshape(-1)), d(lab[4].reshape(-1)), use_graph=False)
    torch.cuda.synchronize()
    ref = net.params.cpu().numpy()
    assert not np.array_equal(ref, start)
    # same arithmetic up to the fp32 summation order of the two shard gradients; Adam's sign-like first steps
    # amplify ~eps gradients (cf. test_ppo_train_matches_oracle_and_graph_replay_is_bitwise: 5e-3 over 6 steps)
    assert _delta_err(got, ref, start) < 5e-3, _delta_err(got, ref, start)
    assert np.abs(got - ref).max() <= 2 * 6 * cfg["LR"]


def test_weak_scaling_two_ranks_equals_one_process_on_the_union_minibatch(tmp_path):
    """Weak mode: rank r owns its own 80-row rollout and full 32-row local minibatches, grad_scale 1/2 -> the update
    of ONE process whose minibatch is the union (64 rows: the means of two equal halves average to the mean of the
    union)."""
    import dp_worker
    from xingtian_amd.model.hip_net import HipActorCritic
    got = _run_two_ranks(tmp_path, "weak")
    spec, cfg, n = dp_worker.ppo_case()
    rolls = [dp_worker.ppo_rollout(200 + r, n) for r in range(2)]
    _, _, perms = dp_worker.ppo_rollout(100, n)
    obs = np.concatenate([r[0] for r in rolls])
    lab = [np.concatenate([r[1][i] for r in rolls]) for i in range(5)]
    # minibatch k of the union = rank 0's rows of its minibatch k followed by rank 1's
    b = cfg["BATCH_SIZE"]
    uperm = []
    for ep in range(2):
        row = []
        for s in range(0, n, b):
            row += list(perms[ep, s:s + b]) + list(n + perms[ep, s:s + b])
        uperm.append(row)
    uperm = np.asarray(uperm, np.int32)
    cfg2 = dict(cfg, BATCH_SIZE=2 * b)
    net = HipActorCritic(spec, max_batch=2 * b, seed=5)
    start = net.params.cpu().numpy().copy()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    net.ppo_train(net.make_ppo_cfg(cfg2), net.to_device_obs(obs), d(uperm), d(lab[0]), d(lab[1].reshape(-1)),
                  d(lab[2].reshape(-1)), d(lab[3].reshape(-1)), d(lab[4].reshape(-1)), use_graph=False)
    torch.cuda.synchronize()
    ref = net.params.cpu().numpy()
    assert _delta_err(got, ref, start) < 5e-3, _delta_err(got, ref, start)


def test_impala_sum_loss_two_ranks_whole_trajectory_shards(tmp_path):
    """IMPALA's loss is a SUM over (T-1) x B (impala_cnn_opt.py:299-318,351): 5 trajectories as shards of 3 + 2,
    gradients summed without scaling, two updates == the single-process updates on all 5 trajectories."""
    import dp_worker
    from xingtian_amd.model.hip_net import HipActorCritic
    got = _run_two_ranks(tmp_path, "impala")
    spec, data, tlen, ntraj = dp_worker.impala_case()
    net = HipActorCritic(spec, max_batch=tlen * ntraj, seed=5)
    start = net.params.cpu().numpy().copy()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    c = net.make_impala_cfg(1e-3, 40.0, tlen)
    for _ in range(2):
        net.impala_step(c, d(data["obs"]), d(data["bp"]), d(data["act"]), d(data["done"]), d(data["rew"]), apply=True)
    torch.cuda.synchronize()
    ref = net.params.cpu().numpy()
    assert _delta_err(got, ref, start) < 5e-3, _delta_err(got, ref, start)


def test_bench_self_spawns_two_ranks_and_reports_weak_strict_and_impala():
    """``python bench.py --gpus 2`` outside a launcher must start the two ranks itself and print ONE JSON line with
    ``n_gpus: 2``, the weak-scaling headline, the strict-sharding result and the IMPALA data-parallel secondaries
    (here with the diagnostic gloo transport, the two ranks sharing the one GPU)."""
    import json
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix="xt_bench_"), "detail.json")     # (not the tree's bench_detail.json)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--test-backend", "gloo", "--steps", "2", "--warmup", "1",
           "--detail-file", detail]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    lines = [l for l in proc.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["ranks_in_group"] == 2
    assert d["config"]["global_batch"] == 640 and d["config"]["env_steps_per_update"] == 8192
    assert len(lines[0]) < 4000                      # the driver-facing line stays compact (BENCH_r04: parsed = null)
    assert d["strict"]["rows_per_gpu"] == 160 and d["strict"]["value"] > 0
    # the N > 1 line is a complete line: roofline + cpu_baseline, and the strict (reference-semantics)
    # number sits at the top level next to the weak-mode `value`
    assert d["roofline"]["frac"] > 0 and d["roofline"]["bound"] == "mfma" and d["cpu_baseline"]["value"] > 0
    assert d["value_strict"] == d["strict"]["value"] and d["global_batch"] == 640 and d["global_batch_strict"] == 320
    assert d["secondary"]["pong_impala_speedup"]["strict"] > 0 and d["secondary"]["breakout_impala"]["weak"] > 0
    # the direct all-reduce over hipIpc-mapped memory carries a validated number even with both ranks on one GPU (the RCCL
    # variants cannot form a communicator there and say so)
    assert isinstance(d["dp_variants"]["direct"], float) and d["dp_variants"]["direct"] > 0, d["dp_variants"]
    assert d["detail"] == detail
    full = json.load(open(detail))
    assert full["strict"]["global_batch"] == 320 and full["strict"]["dp_variants"]["direct"]["valid"], full["strict"]
    assert full["dp_variants"]["direct"]["first_update_bitwise_equal"] is True
    sec = {s["workload"].split()[0]: s for s in full["secondary"]}
    assert sec["examples/pong_impala_speedup.yaml"]["strict"]["trajectories_per_rank"] == "10..10"
    assert sec["examples/breakout_impala.yaml"]["weak"]["frames_per_train_global"] == 256
    # a launcher environment with another world size is refused, never silently reduced
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--quick"],
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=120)
    assert bad.returncode != 0 and b"WORLD_SIZE=2" in bad.stderr


def test_exchange_hook_with_two_ranks_matches_the_stepwise_path_bitwise(tmp_path):
    """The library's gradient-exchange hook (``xt_net_ppo_train`` enqueues fwd/bwd -> hook -> norm/clip/Adam itself) met
    by two real ranks, with one bucket and with XT_XCHG_OVERLAP's two buckets (last trunk layer + heads exchanged from
    the side stream right after the first backward launch, the conv layers afterwards): both reproduce the step-wise
    weak-mode update bit for bit -- a 2-rank sum is the same in any bucket split -- and the hook is called with exactly
    the documented sub-ranges, in the same order on both ranks."""
    ref = _run_two_ranks(tmp_path, "weak")
    for mode in ("hook", "hook_overlap"):
        got = _run_two_ranks(tmp_path, mode)
        assert np.array_equal(got, ref), mode


def test_strict_sharding_inside_the_library_call_matches_the_stepwise_strict_path_bitwise(tmp_path):
    """xt_ppo_cfg.shard_rank / shard_world (ABI 9): ONE xt_net_ppo_train call per rank walks the shared permutations,
    takes this rank's balanced shard of every global minibatch (16 + 16 rows, 8 + 8 in the short last one), exchanges
    through the hook and applies clip + Adam to the exchanged gradient -- bit for bit the Python-stepped strict path
    (and therefore the single-GPU update up to fp32 summation order, test_strict_sharding_two_ranks_...)."""
    ref = _run_two_ranks(tmp_path, "strict")
    got = _run_two_ranks(tmp_path, "strict_hook")
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("mode", ["impala_rms", "impala_sched"])
def test_data_parallel_impala_with_rmsprop_and_scheduled_step_sizes(tmp_path, mode):
    """impala_cnn_opt.py:198-217,234-249 data parallel: opt_type rmsprop (centred RMSProp on the EXCHANGED gradient)
    and an lr_schedule step size read from device memory, 5 trajectories as shards of 3 + 2, against the single-process
    xt_net_impala_train with the same optimiser settings."""
    import dp_worker
    from xingtian_amd.model.hip_net import HipActorCritic
    got = _run_two_ranks(tmp_path, mode)
    spec, data, tlen, ntraj = dp_worker.impala_case()
    net = HipActorCritic(spec, max_batch=tlen * ntraj, seed=5)
    start = net.params.cpu().numpy().copy()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if mode == "impala_rms":
        net.set_optimizer("rmsprop")
    c = net.make_impala_cfg(1e-3, 40.0, tlen, opt_type="rmsprop" if mode == "impala_rms" else "adam")
    for step in range(2):
        lr_steps = d(np.asarray([7e-4 / (step + 1)], np.float32)) if mode == "impala_sched" else None
        net.impala_train(c, d(data["obs"]), tlen * ntraj, d(data["bp"]), d(data["act"]), d(data["done"]), d(data["rew"]),
                         lr_steps=lr_steps, use_graph=False)
    torch.cuda.synchronize()
    ref = net.params.cpu().numpy()
    assert not np.array_equal(ref, start)
    assert _delta_err(got, ref, start) < 5e-3, _delta_err(got, ref, start)


@pytest.mark.parametrize("mode", ["impala_one_traj", "impala_rms_one_traj"])
def test_data_parallel_impala_with_an_empty_shard(tmp_path, mode):
    """fewer trajectories than ranks: the empty rank contributes a zero gradient in BOTH branches of dp_impala_step (the
    step-wise Adam path and the hook path with centred RMSProp) -- same contract, same result as one process."""
    import dp_worker
    from xingtian_amd.model.hip_net import HipActorCritic
    got = _run_two_ranks(tmp_path, mode)
    spec, data, tlen, _ = dp_worker.impala_case()
    net = HipActorCritic(spec, max_batch=tlen, seed=5)
    start = net.params.cpu().numpy().copy()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rms = mode == "impala_rms_one_traj"
    if rms:
        net.set_optimizer("rmsprop")
    c = net.make_impala_cfg(1e-3, 40.0, tlen, opt_type="rmsprop" if rms else "adam")
    sl = slice(0, tlen)
    for _ in range(2):
        net.impala_train(c, d(data["obs"][sl]), tlen, d(data["bp"][sl]), d(data["act"][sl]), d(data["done"][sl]),
                         d(data["rew"][sl]), use_graph=False)
    torch.cuda.synchronize()
    ref = net.params.cpu().numpy()
    assert not np.array_equal(ref, start)
    assert _delta_err(got, ref, start) < 5e-3, _delta_err(got, ref, start)
