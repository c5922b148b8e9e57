#!/usr/bin/env python
"""Learner-throughput benchmark of the MI355X-native PPO update (BASELINE.json metric).

One "step" = one complete learner update on one synthetic rollout batch that is already
resident in HBM: GAE over env_num x T transitions (xt/agent/ppo/ppo.py:77-106 moved to the
learner) followed by Model.train (xt/model/ppo/ppo.py:111-132): NUM_SGD_ITER epochs x
ceil(N/BATCH_SIZE) minibatch SGD steps (forward, PPO loss, backward, global-norm clip,
Adam).  Workload = BASELINE.json configs[1]: examples/breakout_ppo.yaml, PpoCnn 84x84x4,
env_num=32, T=128 (N=4096 samples), BATCH_SIZE=320, NUM_SGD_ITER=4, hidden 256, A=4.

value = env-frames/s = 4 (frame-skip) x env-steps consumed / wall time, whole job over all
ranks.  N>1: one process per GPU (torchrun), weak scaling: every rank owns env_num=32
trajectories and a 320-row local minibatch; gradients are summed with one RCCL all-reduce of
the flat fp32 gradient buffer per SGD step, then every rank applies the identical
clip+Adam update (grad_scale = 1/N).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
FRAME_SKIP = 4                  # xt/environment/gym/atari_wrappers.py:34

CFG = dict(LR=2.5e-4, LOSS_CLIPPING=0.1, ENTROPY_LOSS=0.003, VF_CLIP=5.0, CRITIC_LOSS_COEF=1.0,
           MAX_GRAD_NORM=5.0, BATCH_SIZE=320, NUM_SGD_ITER=4)
ENV_NUM, T_LEN, STATE_DIM, A_DIM, HIDDEN = 32, 128, (84, 84, 4), 4, (256,)


def synth_rollout(seed, env_num=ENV_NUM, t_len=T_LEN):
    rng = np.random.default_rng(seed)
    n = env_num * t_len
    obs = rng.integers(0, 256, (n,) + STATE_DIM, dtype=np.uint8)
    action = rng.integers(0, A_DIM, n).astype(np.int32)
    logits = rng.standard_normal((n, A_DIM))
    lsm = logits - np.log(np.exp(logits).sum(-1, keepdims=True))
    logp = np.take_along_axis(lsm, action[:, None].astype(np.int64), 1).astype(np.float32).reshape(-1)
    value = rng.standard_normal((env_num, t_len + 1)).astype(np.float32)
    reward = rng.choice([-1.0, 0.0, 1.0], size=(env_num, t_len), p=[0.05, 0.9, 0.05])
    done = (rng.random((env_num, t_len)) < 0.01)
    return obs, action, logp, value, reward, done


def host_cores():
    """cores this process may actually use: min(cpu_count, affinity mask, cgroup cpu quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(obs, action, logp, value, reward, done, max_seconds=20.0):
    """Oracle (torch-CPU fp32 restatement of the same update, oneDNN, all host cores) timed on a
    bounded sample: a few B=320 SGD steps + the numpy GAE of the full rollout; extrapolated to the
    full update.  Checker/baseline only -- never on the product path."""
    from oracle import nets, returns, torch_ref
    cores = min(host_cores(), 64)   # oneDNN does not scale past ~64 threads at B=320; count is reported
    torch.set_num_threads(cores)
    spec = nets.ppo_cnn_spec(STATE_DIM, A_DIM, HIDDEN, "relu", True)
    params = nets.init_params(spec, seed=0)
    learner = torch_ref.TorchPpoLearner(spec, params, CFG, torch.float32)
    t0 = time.perf_counter()
    advs = []
    for i in range(value.shape[0]):
        a, _, tg = returns.gae(value[i].reshape(-1, 1), reward[i].copy(), done[i])
        advs.append((a, tg))
    t_gae = time.perf_counter() - t0
    n = obs.shape[0]
    b = CFG["BATCH_SIZE"]
    adv = np.concatenate([a for a, _ in advs]).astype(np.float32)
    tgt = np.concatenate([t for _, t in advs]).astype(np.float32)
    oldv = value[:, :-1].reshape(-1, 1)
    rng = np.random.default_rng(0)
    steps, t_steps = 0, 0.0
    learner.step(obs[:b], action[:b], logp[:b].reshape(-1, 1), adv[:b], oldv[:b], tgt[:b])   # warm-up
    while t_steps < max_seconds and steps < 24:
        mb = rng.permutation(n)[:b]
        t1 = time.perf_counter()
        learner.step(obs[mb], action[mb], logp[mb].reshape(-1, 1), adv[mb], oldv[mb], tgt[mb])
        t_steps += time.perf_counter() - t1
        steps += 1
    per_step = t_steps / steps
    nsteps_full = CFG["NUM_SGD_ITER"] * ((n + b - 1) // b)
    t_full = t_gae + per_step * nsteps_full
    return {"value": FRAME_SKIP * n / t_full, "unit": "env-frames/s", "cores": cores, "kind": "port",
            "sample": "{} SGD steps of B={} (fp32 torch-CPU restatement, oneDNN) + numpy GAE of {}x{}; "
                      "extrapolated to {} steps/update".format(steps, b, value.shape[0], T_LEN, nsteps_full),
            "ms_per_sgd_step": per_step * 1e3}


def main_impala(args):
    """Secondary workload: examples/breakout_impala.yaml -- one 'step' = 64 learner trains, each on one
    128-frame message (prepare_times_per_train=1, BATCH_SIZE 512 >= 128), rollout resident in HBM."""
    if int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise RuntimeError("--workload impala is a single-GPU measurement")
    torch.cuda.set_device(0)
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    n_msg, t_len, a_dim = 64, 128, 4
    rng = np.random.default_rng(0)
    n = n_msg * t_len
    dev = torch.device("cuda", 0)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    obs = d(rng.integers(0, 256, (n,) + STATE_DIM, dtype=np.uint8))
    bp = d(rng.standard_normal((n, a_dim)).astype(np.float32))
    act = d(rng.integers(0, a_dim, n).astype(np.int32))
    done = d((rng.random(n) < 0.01).astype(np.uint8))
    rew = d(rng.choice([-1.0, 0.0, 1.0], n, p=[0.05, 0.9, 0.05]).astype(np.float32))
    spec = netspec.impala_cnn_opt(STATE_DIM, a_dim, 0.0, 255.0)
    net = HipActorCritic(spec, max_batch=t_len, seed=0)
    cfg = net.make_impala_cfg(5e-4, 40.0, t_len)

    def one_update():
        for i in range(n_msg):
            sl = slice(i * t_len, (i + 1) * t_len)
            net.impala_step(cfg, obs[sl], bp[sl], act[sl], done[sl], rew[sl], apply=True)

    for _ in range(args.warmup):
        one_update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_update()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert torch.isfinite(net.params).all()
    _emit({
        "metric": "learner env-frames/sec (Atari 84x84x4)", "value": FRAME_SKIP * n * args.steps / el,
        "unit": "env-frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp32", "data": "synthetic",
        "config": {"workload": "examples/breakout_impala.yaml ImpalaCnnOpt 84x84x4 uint8 + v-trace, 64 messages x "
                               "T=128 frames per step (one SGD step per message), HBM-resident", "parallelism": "dp1"}})


_REAL_STDOUT = None


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries print there too (RCCL's version banner at
    communicator creation), so file descriptor 1 is pointed at stderr for the whole run and the result line goes
    to a private duplicate of the original stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _REAL_STDOUT


def _emit(obj):
    out = _claim_stdout()
    out.write(json.dumps(obj) + "\n")
    out.flush()


def main():
    _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--force-dp-path", action="store_true",
                    help="run the N>1 code path (step-wise fwd/bwd -> RCCL all-reduce -> clip+Adam) even with one rank: "
                         "validates the data-parallel plumbing on a single GPU")
    ap.add_argument("--dp-mode", default="eager", choices=["eager", "eager-overlap", "graph", "graph-overlap", "ingraph"],
                    help="data-parallel path (N>1 or --force-dp-path).  eager (default): step-wise fwd/bwd -> one RCCL "
                         "all-reduce of the flat gradient -> clip+Adam.  *overlap: two buckets, the Dense+heads gradient "
                         "(95 %% of the bytes) all-reduced asynchronously under the conv backward.  graph*: the compute "
                         "segments replayed from hipGraphs (parallel.DpGraphStepper).  Measured with a 1-rank RCCL group "
                         "(ms per update): eager 8.4, graph 9.4, eager-overlap 11.4, graph-overlap 12.2 -- three small graph "
                         "launches cost more than the dozen eager launches they replace, and every extra c10d call with "
                         "its cross-stream events about 25 us; the alternatives are kept for interconnects where the "
                         "all-reduce itself is the larger term.  ingraph: raw RCCL all-reduces enqueued by the library "
                         "itself (xt_net_set_grad_exchange) and captured into the hipGraph of the whole update -- no "
                         "host involvement per step; validated on one rank only, hence opt-in")
    ap.add_argument("--workload", default="ppo", choices=["ppo", "impala"],
                    help="ppo = BASELINE configs[1] (the headline metric, default); impala = configs[2] "
                         "(breakout_impala.yaml, ImpalaCnnOpt + v-trace, env_num=64 messages of T=128), secondary")
    args = ap.parse_args()
    if args.workload == "impala":
        return main_impala(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the learner path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dp_path:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torchrun --nproc-per-node {}".format(args.gpus)

    from xingtian_amd import lib as L
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    from xingtian_amd.parallel import DpGraphStepper, RcclComm, dp_ppo_step

    dev = torch.device("cuda", local_rank)
    obs, action, logp, value, reward, done = synth_rollout(seed=rank)
    n = obs.shape[0]
    spec = netspec.ppo_cnn(STATE_DIM, A_DIM, HIDDEN, "relu", True)
    net = HipActorCritic(spec, max_batch=CFG["BATCH_SIZE"], device=str(dev), seed=0)   # same seed -> same replica
    lib = L.load()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_obs, d_act, d_logp = d(obs), d(action), d(logp)
    d_value, d_reward, d_done = d(value), d(reward), d(done.astype(np.uint8))
    d_adv = torch.empty((n,), dtype=torch.float64, device=dev)
    d_tgt = torch.empty((n,), dtype=torch.float64, device=dev)
    d_oldv = torch.empty((n,), dtype=torch.float32, device=dev)
    perm_rng = np.random.default_rng(1234)   # identical on every rank
    d_perm = torch.empty((CFG["NUM_SGD_ITER"], n), dtype=torch.int32, device=dev)
    cfg = net.make_ppo_cfg(CFG, grad_scale=1.0 / world, global_batch=0)
    dp_path = world > 1 or args.force_dp_path
    use_graph = (not dp_path) and not args.no_graph
    bsz = CFG["BATCH_SIZE"]

    def new_perms():
        inds = np.arange(n)
        p = np.empty((CFG["NUM_SGD_ITER"], n), np.int32)
        for ep in range(CFG["NUM_SGD_ITER"]):
            perm_rng.shuffle(inds)
            p[ep] = inds
        d_perm.copy_(torch.from_numpy(p), non_blocking=False)

    rccl = None
    if dp_path and args.dp_mode == "ingraph":
        rccl = RcclComm(rank, world)
        rccl.all_reduce_(net.grads.zero_(), L.stream_ptr())      # RCCL's lazy set-up outside any capture
        torch.cuda.synchronize()
        rccl.attach(net)
    stepper = None
    if dp_path and args.dp_mode.startswith("graph"):
        stepper = DpGraphStepper(net, cfg, CFG["LR"], CFG["MAX_GRAD_NORM"], d_obs, d_act, d_logp, d_adv, d_oldv, d_tgt,
                                 world, overlap=(args.dp_mode == "graph-overlap"))

    def one_update():
        new_perms()
        st = L.stream_ptr()
        L.check(lib.xt_gae_f64(L.ptr(d_value), L.ptr(d_reward), L.ptr(d_done), L.ptr(d_adv), L.ptr(d_tgt),
                               L.ptr(d_oldv), ENV_NUM, T_LEN, 0.99, 0.95, st), "gae")
        if not dp_path or rccl is not None:
            net.ppo_train(cfg, d_obs, d_perm, d_act, d_logp, d_adv, d_oldv, d_tgt,
                          use_graph=(use_graph or rccl is not None) and not args.no_graph)
        else:
            for ep in range(CFG["NUM_SGD_ITER"]):
                for start in range(0, n, bsz):
                    idx = d_perm[ep, start:start + bsz]
                    # fwd/bwd -> RCCL sum over xGMI (Dense+heads bucket overlapped with the conv backward) -> clip+Adam
                    if stepper is not None:
                        stepper.step(idx)
                    else:
                        dp_ppo_step(net, cfg, CFG["LR"], CFG["MAX_GRAD_NORM"], d_obs, idx, d_act, d_logp, d_adv, d_oldv,
                                    d_tgt, world, overlap=(args.dp_mode == "eager-overlap"))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_update()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_update()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(net.params).all(), "non-finite parameters after the benchmark"

    frames = FRAME_SKIP * n * world * args.steps
    out = {
        "metric": "learner env-frames/sec (Atari 84x84x4)", "value": frames / elapsed, "unit": "env-frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp32", "data": "synthetic",
        "config": {"workload": "examples/breakout_ppo.yaml PpoCnn 84x84x4 uint8, env_num=32/GPU, T=128, "
                               "BATCH_SIZE=320/GPU, NUM_SGD_ITER=4, hidden 256, A=4; step = GAE + full PPO update "
                               "(52 SGD steps) on an HBM-resident rollout",
                   "env_steps_per_update": n * world, "sgd_steps_per_update": CFG["NUM_SGD_ITER"] * ((n + bsz - 1) // bsz),
                   "parallelism": "dp{}".format(world), "hip_graph": bool((use_graph or rccl is not None) and not args.no_graph)},
    }
    if rank == 0:
        # ---- roofline of the dominant kernel: every layer kernel of one SGD step timed live with HIP events on
        # the launch stream (xt_net_time_layer), algorithmic FLOPs = 2*M*N*K per GEMM
        idx = d_perm[0, :bsz].contiguous()
        kern = {}
        for li, lay in enumerate(spec.layers):
            mnk2 = 2.0 * bsz * lay.OH * lay.OW * lay.N * lay.K
            if li == 0:
                kern["L0 conv8x8/4 fwd  [conv_u8c4k8_fwd_flat_kernel]"] = (net.time_layer(0, 0, d_obs, idx, bsz, 50), mnk2, "bf16x3")
                kern["L0 conv8x8/4 wgrad [conv_u8c4k8_wgrad_flat_kernel]"] = (net.time_layer(0, 1, d_obs, idx, bsz, 50), mnk2, "bf16x3")
            else:
                kern["L%d %s fwd  [igemm_fwd_kernel | direct_fwd_kernel]" % (li, lay.name)] = (net.time_layer(li, 0, d_obs, idx, bsz, 50), mnk2, "fp32")
                # conv backward launches: the input-gradient half runs bf16x6 (six bf16 MFMAs per 16-deep chunk), the
                # weight-gradient half fp32 MFMA -> peak of the launch = harmonic mean of the two halves' peaks
                x6 = os.environ.get("XT_BF16X6", "1") != "0" and lay.KH > 1
                kern["L%d %s dgrad+wgrad [igemm_bwd_layer_kernel]" % (li, lay.name)] = (
                    net.time_layer(li, 3, d_obs, idx, bsz, 50), 2 * mnk2, "fp32+bf16x6" if x6 else "fp32")
        dom = max(kern, key=lambda k: kern[k][0])
        ms, flops, kind = kern[dom]
        # fp32 kernels: v_mfma_f32_32x32x2_f32 dense peak; bf16x3 kernels spend 3 bf16 MFMA flops per algorithmic flop
        peak = {"fp32": FP32_MFMA_PEAK_TFLOPS, "bf16x3": 2500.0 / 3.0,
                "fp32+bf16x6": 2.0 / (1.0 / FP32_MFMA_PEAK_TFLOPS + 6.0 / 2500.0)}[kind]
        ach = flops / (ms * 1e-3) / 1e12
        # HBM traffic of the dominant kernel: PMC FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE per launch from the
        # committed rocprofv3 passes (profiles/r01_pmc_traffic.json holds every layer kernel), matched by kernel name
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        match = {"L0 conv8x8/4 fwd": "conv_u8c4k8_fwd", "L0 conv8x8/4 wgrad": "conv_u8c4k8_wgrad",
                 "L1 shared_conv_layer_1 fwd": "direct_fwd_kernel", "L2 shared_conv_layer_2 fwd": "igemm_fwd_kernel<64, 64",
                 "L3 shared_hidden_mlp_0 fwd": "igemm_fwd_kernel<64, 64",
                 "L1 shared_conv_layer_1 dgrad+wgrad": "igemm_bwd_layer_kernel<128, 32, 4, 1, false, 128, 32, 4, 1",
                 "L2 shared_conv_layer_2 dgrad+wgrad": "igemm_bwd_layer_kernel<64, 64, 2, 2, false, 128, 32, 4, 1",
                 "L3 shared_hidden_mlp_0 dgrad+wgrad": "igemm_bwd_layer_kernel<64, 64, 2, 2, false, 64, 64, 2, 2"}
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                sub = next((v for k, v in match.items() if dom.startswith(k)), None)
                for row in tj.get("all_kernels_KB", []):
                    if sub and sub in row["kernel"]:
                        traffic = (2.0 * row["FETCH_SIZE_KB"] + row["WRITE_SIZE_KB"]) * 1024.0
                        break
            except (OSError, ValueError, KeyError):
                traffic = None
        out["roofline"] = {"bound": "mfma", "kernel": dom, "arith": kind, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                           "frac": ach / peak, "traffic": traffic, "flop_per_launch": flops, "avg_launch_ms": ms,
                           "kernels_us": {k: round(v[0] * 1e3, 2) for k, v in kern.items()},
                           "sum_layer_kernels_us": round(sum(v[0] for v in kern.values()) * 1e3, 1)}
        total_flops = 31.313e6 * n * CFG["NUM_SGD_ITER"]
        out["update_tflops"] = total_flops * args.steps / elapsed / 1e12
        out["config"]["dp_path"] = bool(dp_path)
        if dp_path:
            out["config"]["dp_mode"] = args.dp_mode if not (stepper is not None and stepper.failed) else "eager (capture failed)"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(obs, action, logp, value, reward, done)
        _emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
