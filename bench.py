#!/usr/bin/env python
"""Learner-throughput benchmark of the MI355X-native PPO / IMPALA update (BASELINE.json metric).

Headline (``value``): BASELINE.json configs[1] = examples/breakout_ppo.yaml, PpoCnn 84x84x4 uint8, env_num=32,
T=128 (N=4096 samples), BATCH_SIZE=320, NUM_SGD_ITER=4, hidden 256, A=4.  One "step" = one complete learner update
on one synthetic rollout that is already resident in HBM: GAE over env_num x T transitions
(xt/agent/ppo/ppo.py:77-106 moved to the learner) + Model.train (xt/model/ppo/ppo.py:111-132): 52 minibatch SGD
steps (forward, PPO loss, backward, global-norm clip, Adam).  value = env-frames/s = 4 (frame-skip) x env-steps
consumed / wall time, whole job over all ranks.

The same JSON line also carries (rank 0, one GPU):
  roofline      dominant kernel of an SGD step, timed live with HIP events on the launch stream
  e2e           SURVEY 8(d)'s metric through the PLUGIN classes: alg.prepare_data x env_num (host numpy ->
                pinned staging -> async H2D) + alg.train() + alg.get_weights() (D2H), env_num 32 and the YAML's 10
  secondary     configs[2] breakout_impala.yaml (ImpalaCnnOpt 84x84, T=128, one SGD step per message) and
                configs[4] pong_impala_speedup.yaml (42x42, A=6, 1000-frame steps), HBM-resident and through
                IMPALAOpt, each with its own roofline and cpu_baseline
  cpu_baseline  the torch-CPU fp32 restatement of the same update on the host cores (all cores and 1 thread) and the
                numpy GAE per trajectory (bit-for-bit the reference's PPO.data_proc)
  sustained     the headline loop repeated for >= 2 s
  device        diagnostic: host CPU model and the GPU's clocks / power (rocm-smi) sampled WHILE the headline loop runs,
                after the timed region (boxes of one pool have measured 7.0 and 11 ms for the same update)

N > 1: ``python bench.py --gpus N`` spawns N ranks itself (torch.distributed.run, one process per GPU, RCCL) when it
is not already running under a launcher, and FAILS if it cannot.  Two data-parallel modes are measured in the same
run: ``value`` = weak scaling (every rank owns env_num=32 trajectories and a 320-row local minibatch, one all-reduce
of the flat fp32 gradient per SGD step, grad_scale 1/N); ``strict`` = the reference's global minibatch of 320 rows
split into N shards (40 rows per GPU at N=8), shared permutation, means over the global minibatch.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16
FRAME_SKIP = 4                  # xt/environment/gym/atari_wrappers.py:34

CFG = dict(LR=2.5e-4, LOSS_CLIPPING=0.1, ENTROPY_LOSS=0.003, VF_CLIP=5.0, CRITIC_LOSS_COEF=1.0,
           MAX_GRAD_NORM=5.0, BATCH_SIZE=320, NUM_SGD_ITER=4)
ENV_NUM, T_LEN, STATE_DIM, A_DIM, HIDDEN = 32, 128, (84, 84, 4), 4, (256,)
PPO_MFLOP_PER_SAMPLE_PASS = 31.313
IMPALA = {   # MFLOP per sample (fwd + bwd, one pass)
    "breakout_impala": dict(dim=84, a_dim=4, t_len=128, frames_per_train=128, mean=0.0, std=255.0, lr=5e-4,
                            mflop=19.128, trains=64,
                            name="examples/breakout_impala.yaml ImpalaCnnOpt 84x84x4 uint8 + v-trace, T=128, "
                                 "vector_env_size=1, prepare_times_per_train=1 (one 128-frame SGD step per message)"),
    "pong_impala_speedup": dict(dim=42, a_dim=6, t_len=50, frames_per_train=1000, mean=128.0, std=128.0, lr=1e-3,
                                mflop=13.712, trains=16, train_per_checkpoint=3,     # pong_impala_speedup.yaml:5
                                name="examples/pong_impala_speedup.yaml ImpalaCnnOpt 42x42x4 uint8 (mean 128 / std 128) "
                                     "A=6, T=50, 4 messages x 5 envs per train (one 1000-frame SGD step)"),    # SURVEY 8(d) C3 variants (flagged): the YAML's BATCH_SIZE = 512 filled with 4 messages per SGD step instead of
    # prepare_times_per_train = 1 (a semantic change: 4x the data per optimiser step), and pong trained per rollout
    # message (5 envs x 50 steps = 250 frames per SGD step) instead of per 4 messages
    "breakout_impala_batched": dict(dim=84, a_dim=4, t_len=128, frames_per_train=512, mean=0.0, std=255.0, lr=5e-4,
                                    mflop=19.128, trains=16, semantic_change=True, msgs_per_train=4,
                                    name="examples/breakout_impala.yaml:5-6 BATCH_SIZE=512 filled: 4 messages x T=128 per "
                                         "SGD step (prepare_times_per_train 1 -> 4: semantic change, flagged)"),
    "pong_impala_per_message": dict(dim=42, a_dim=6, t_len=50, frames_per_train=250, mean=128.0, std=128.0, lr=1e-3,
                                    mflop=13.712, trains=64, semantic_change=True, msgs_per_train=1,
                                    name="examples/pong_impala_speedup.yaml trained per rollout message: 5 envs x T=50 = 250 "
                                         "frames per SGD step (prepare_times_per_train 4 -> 1: semantic change, flagged)"),
}


# ------------------------------------------------------------------------------------------------ plumbing
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


# ------------------------------------------------------------------------------------------------ the ONE compact line
COMPACT_LIMIT = 4000    # bytes; round 4's 25.7 KB line could not be parsed by the driver (BENCH_r04.json: parsed = null)
DETAIL_NAME = "bench_detail.json"
DTYPE = "fp32 via split-bf16 MFMA (bf16x3/x6, fp32 accumulate); weight gradients bf16x6 or fp32 MFMA per `roofline.arith`"
PARITY = ("GAE + alg host logic + loss/dist/v-trace formulas: executed-reference goldens (tests/golden); "
          "TF-library semantics (Conv2D/Dense grads, Adam, clip_by_global_norm): restated fp64, unpinned (no TF here)")


def _r(x, nd=4):
    """round floats for the compact line (significant digits, not decimals)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float("%.*g" % (nd + 2, float(x)))
    except (TypeError, ValueError):
        return x


def _pick(src, keys, nd=4):
    return {k: _r(src[k], nd) for k in keys if isinstance(src, dict) and k in src}


def compact_line(out):
    """The driver-facing line: the contract fields + `roofline` + `cpu_baseline` + SURVEY 8(d)'s `value_e2e` with its
    three components + one summary number per secondary workload; everything else lives in bench_detail.json.  Pure
    function of the full result dict (tests/test_cpu_host.py builds it from a canned result and bounds its length)."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline"), 6)
    line["dtype"] = DTYPE
    line["data"] = out.get("data", "synthetic")
    cfg = out.get("config", {})
    line["config"] = _pick(cfg, ("workload", "env_steps_per_update", "sgd_steps_per_update", "global_batch", "parallelism",
                                 "hip_graph", "dp_mode", "ranks_in_group", "DIAGNOSTIC"))
    if isinstance(line["config"].get("workload"), str):
        line["config"]["workload"] = line["config"]["workload"][:320]
    line["value_includes_h2d"] = False
    line["update_tflops"] = _r(out.get("update_tflops"))
    roof = out.get("roofline")
    if roof:
        line["roofline"] = _pick(roof, ("bound", "kernel", "kernel_symbol", "arith", "achieved", "peak", "unit", "frac", "traffic",
                                        "mfma_pipe_util_pmc", "pmc_source", "avg_launch_ms", "flop_per_launch"))
        sym = line["roofline"].get("kernel_symbol")
        if isinstance(sym, str) and len(sym) > 80:
            line["roofline"]["kernel_symbol"] = sym[:80]
        line["roofline"]["timing"] = "in-graph" if str(roof.get("timing", "")).startswith("in-graph") else "isolated"
    cpu = out.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = _pick(cpu, ("value", "unit", "cores", "kind", "ms_per_sgd_step", "spread", "host_cpu"))
        line["cpu_baseline"]["sample"] = str(cpu.get("sample", ""))[:140]
    e2e = (out.get("e2e") or {})
    first = e2e.get("env_num_32_pinned_ring") if isinstance(e2e.get("env_num_32_pinned_ring"), dict) and \
        "value" in e2e.get("env_num_32_pinned_ring", {}) else e2e.get("env_num_32")
    if isinstance(first, dict) and "value" in first:
        line["value_e2e"] = _r(first["value"], 6)
        line["e2e"] = {"includes": "prepare_data x env_num (H2D of the uint8 rollout) + train() + weights hand-over (D2H)",
                       "ms_per_update": _r(first.get("ms_per_update")), "prepare_data_ms": _r(first.get("prepare_data_ms")),
                       "train_ms": _r(first.get("train_ms")), "weights_ms": _r(first.get("get_weights_ms")),
                       "host_path": "pinned ShmRing + publish_weights" if first is e2e.get("env_num_32_pinned_ring")
                       else "pageable numpy + get_weights()"}
        alt = e2e.get("env_num_32")
        if first is not alt and isinstance(alt, dict) and "value" in alt:
            line["e2e"]["value_pageable_get_weights"] = _r(alt["value"], 6)
    for k in ("env_num_256", "actor_scan"):
        if isinstance(out.get(k), dict):
            line[k] = {kk: _r(v) for kk, v in out[k].items() if not isinstance(v, (dict, list))}
            for kk, v in list(line[k].items()):
                if isinstance(v, str) and len(v) > 100:
                    line[k][kk] = v[:100]
    if isinstance(out.get("sustained"), dict):
        line["sustained"] = _pick(out["sustained"], ("value", "ms_per_step", "seconds"), 6)
    if "degraded_box" in out:
        line["degraded_box"] = out["degraded_box"]
    if isinstance(out.get("library"), dict):
        line["library"] = _pick(out["library"], ("built_from_sources_sha", "stale_binary"))
    sec = {}
    for w in out.get("secondary") or []:
        if isinstance(w, dict) and "value" in w and not w.get("semantic_change"):
            key = "pong_impala_speedup" if "pong" in str(w.get("workload")) else "breakout_impala"
            sec[key] = {"value": _r(w["value"], 6), "us_per_train": _r(w.get("us_per_train")),
                        "roofline_frac": _r((w.get("roofline") or {}).get("frac")),
                        "value_e2e": _r(((w.get("e2e_ring_prefetch") if "value" in (w.get("e2e_ring_prefetch") or {}) else None)
                                         or w.get("e2e_publish") or w.get("e2e") or {}).get("value"), 6),
                        "value_e2e_blocking": _r((w.get("e2e_publish") or w.get("e2e") or {}).get("value"), 6)}
        elif isinstance(w, dict) and ("weak" in w or "strict" in w or "error" in w):
            key = "pong_impala_speedup" if "pong" in str(w.get("workload")) else "breakout_impala"
            sec[key] = {m: _r(w[m].get("value"), 6) for m in ("weak", "strict") if isinstance(w.get(m), dict)}
            if "error" in w:
                sec[key]["error"] = str(w["error"])[:120]
    if sec:
        line["secondary"] = sec
    if isinstance(out.get("dp_variants"), dict):
        line["dp_variants"] = {k: (_r(v.get("value"), 6) if v.get("valid") else "invalid: " + str(v.get("error"))[:80])
                               for k, v in out["dp_variants"].items() if isinstance(v, dict)}
    if isinstance(out.get("strict"), dict):
        st = out["strict"]
        line["strict"] = _pick(st, ("value", "ms_per_step", "rows_per_gpu", "scaling"), 6)
        if "error" in st:
            line["strict"]["error"] = str(st["error"])[:120]
        # top level, so that the weak-mode `value` (global minibatch N x BATCH_SIZE: a flagged change of the reference's
        # batch) cannot be mistaken for the reference's semantics: `value_strict` = the reference's GLOBAL minibatch of
        # BATCH_SIZE rows sharded over the ranks (strong scaling), `global_batch` = rows per SGD step behind `value`
        line["value_strict"] = _r(st.get("value"), 6)
        line["global_batch"] = cfg.get("global_batch")
        line["global_batch_strict"] = st.get("global_batch")
    line["parity"] = PARITY
    line["detail"] = out.get("detail_file") or DETAIL_NAME
    txt = json.dumps(line)
    if len(txt) > COMPACT_LIMIT:        # never let the line outgrow the driver again: drop optional blocks, largest first
        for k in ("secondary", "dp_variants", "actor_scan", "env_num_256", "e2e", "sustained", "parity"):
            line.pop(k, None)
            line["truncated"] = True
            if len(json.dumps(line)) <= COMPACT_LIMIT:
                break
    return line


def emit_result(out):
    """full result -> bench_detail.json (next to bench.py, and gpurun_out/ when it exists) + stderr; compact line -> stdout"""
    if out.get("detail_file"):
        paths = [out["detail_file"]]
    else:
        paths = [os.path.join(ROOT, DETAIL_NAME)]
        if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
            paths.append(os.path.join(ROOT, "gpurun_out", DETAIL_NAME))
    for pth in paths:
        try:
            with open(pth, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as exc:
            log("could not write", pth, repr(exc))
    log("detail:", json.dumps(out))
    _emit(compact_line(out))


_T0 = time.perf_counter()


def log(*a):
    print("[bench +{:6.1f}s]".format(time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


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


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()[:60]
    except OSError:
        pass
    return "unknown"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_spawn(args):
    """``python bench.py --gpus N`` outside a launcher: re-exec under torch.distributed.run with N ranks on this
    node.  Never falls back to fewer ranks."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not (args.test_backend and have >= 1):
        raise RuntimeError("bench.py --gpus {} needs {} visible GPUs, found {}".format(args.gpus, args.gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log("spawning", args.gpus, "ranks:", " ".join(cmd))
    proc = subprocess.run(cmd, env=env, stdout=_claim_stdout())
    if proc.returncode != 0:
        raise RuntimeError("bench.py: the {}-rank run failed with exit code {}".format(args.gpus, proc.returncode))


# ------------------------------------------------------------------------------------------------ synthetic data
def synth_rollout(seed, env_num=ENV_NUM, t_len=T_LEN):
    """uint8 frames ~ U{0..255}, actions ~ U{0..3}, logp = log-softmax of N(0,1) logits at the
    action, value ~ N(0,1), reward in {-1,0,1} (p .05/.9/.05), done ~ Bernoulli(0.01)."""
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


def synth_impala(seed, n, dim, a_dim):
    rng = np.random.default_rng(seed)
    return dict(obs=rng.integers(0, 256, (n, dim, dim, 4), dtype=np.uint8),
                logit=rng.standard_normal((n, a_dim)).astype(np.float32),
                action=rng.integers(0, a_dim, n).astype(np.int32),
                done=(rng.random(n) < 0.01),
                reward=rng.choice([-1.0, 0.0, 1.0], n, p=[0.05, 0.9, 0.05]))


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_baseline_ppo(obs, action, logp, value, reward, done, max_seconds=14.0):
    """Oracle (torch-CPU fp32 restatement of the same update, oneDNN) timed on a bounded sample: B=320 SGD steps
    with all host cores and with one thread + the numpy GAE of the full rollout (oracle.returns.gae is bit-for-bit
    the reference's PPO.data_proc, tests/golden/gae_*.npz); extrapolated to the full update.  Checker/baseline only
    -- never on the product path."""
    from oracle import nets, returns, torch_ref
    cores = min(host_cores(), 64)   # oneDNN does not scale past ~64 threads at B=320; the count is reported
    spec = nets.ppo_cnn_spec(STATE_DIM, A_DIM, HIDDEN, "relu", True)
    params = nets.init_params(spec, seed=0)
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
    nsteps_full = CFG["NUM_SGD_ITER"] * ((n + b - 1) // b)

    def timed(threads, budget, max_steps):
        """-> per-step seconds of every timed step (after two untimed warm-up steps)"""
        torch.set_num_threads(threads)
        learner = torch_ref.TorchPpoLearner(spec, params, CFG, torch.float32)
        rng = np.random.default_rng(0)
        for _ in range(2):
            learner.step(obs[:b], action[:b], logp[:b].reshape(-1, 1), adv[:b], oldv[:b], tgt[:b])   # warm-up
        times = []
        while sum(times) < budget and len(times) < max_steps:
            mb = rng.permutation(n)[:b]      # the reference's fancy-index minibatch gather is part of the step
            t1 = time.perf_counter()
            learner.step(obs[mb], action[mb], logp[mb].reshape(-1, 1), adv[mb], oldv[mb], tgt[mb])
            times.append(time.perf_counter() - t1)
        return times

    # (the same code read 6.6 k ... 27.7 k env-frames/s across rounds) the worker threads are pinned to
    # as many CPUs of the allowed set as the process may use (a cgroup quota of 16 on a 256-CPU host let 16 threads
    # wander over all of them and be throttled at random), THREE repeats of 8 steps each are timed, the MEDIAN repeat is
    # reported and the spread with it
    pinned_to = None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        if len(allowed) > cores:
            pinned_to = allowed[:cores]
            os.sched_setaffinity(0, pinned_to)
    except (AttributeError, OSError):
        allowed = None
    try:
        repeats = [timed(cores, max_seconds / 3.0, 8) for _ in range(3)]
        per_repeat = sorted(float(np.median(r)) for r in repeats)
        per_step = per_repeat[1]
        steps = sum(len(r) for r in repeats)
        t1s = timed(1, 6.0, 4)
        per_step_1, steps_1 = float(np.median(t1s)), len(t1s)
    finally:
        if pinned_to is not None:
            try:
                os.sched_setaffinity(0, allowed)
            except OSError:
                pass
    torch.set_num_threads(cores)
    t_full = t_gae + per_step * nsteps_full
    t_full_1 = t_gae + per_step_1 * nsteps_full
    return {"value": FRAME_SKIP * n / t_full, "unit": "env-frames/s", "cores": cores, "kind": "port",
            "sample": "3 repeats x 8 SGD steps of B={} (fp32 torch-CPU restatement, oneDNN; median repeat) + numpy GAE of {}x{}; "
                      "extrapolated to {} steps/update".format(b, value.shape[0], T_LEN, nsteps_full),
            "ms_per_sgd_step": per_step * 1e3,
            "ms_per_sgd_step_repeats": [round(1e3 * x, 3) for x in per_repeat],
            "spread": round((per_repeat[2] - per_repeat[0]) / per_repeat[1], 3),
            "value_range": [FRAME_SKIP * n / (t_gae + per_repeat[2] * nsteps_full), FRAME_SKIP * n / (t_gae + per_repeat[0] * nsteps_full)],
            "steps_timed": steps, "host_cpu": host_cpu_model(), "threads_pinned": bool(pinned_to),
            "one_thread": {"value": FRAME_SKIP * n / t_full_1, "ms_per_sgd_step": per_step_1 * 1e3, "steps": steps_1},
            "gae_numpy_ms_per_trajectory": 1e3 * t_gae / value.shape[0]}


def cpu_baseline_impala(w, data, max_seconds=5.0):
    from oracle import nets, torch_ref
    cores = min(host_cores(), 64)
    torch.set_num_threads(cores)
    spec = nets.impala_cnn_opt_spec((w["dim"], w["dim"], 4), w["a_dim"], w["mean"], w["std"])
    params = nets.init_params(spec, seed=0)
    learner = torch_ref.TorchImpalaLearner(spec, params, dict(LR=w["lr"], grad_norm_clip=40.0,
                                                              sample_batch_step=w["t_len"]), torch.float32)
    f = w["frames_per_train"]
    sl = slice(0, f)
    call = lambda: learner.step(data["obs"][sl], data["logit"][sl], data["action"][sl], data["done"][sl],
                                data["reward"][sl].astype(np.float32))
    call()
    steps, t = 0, 0.0
    while t < max_seconds and steps < 24:
        t1 = time.perf_counter()
        call()
        t += time.perf_counter() - t1
        steps += 1
    return {"value": FRAME_SKIP * f * steps / t, "unit": "env-frames/s", "cores": cores, "kind": "port",
            "sample": "{} SGD steps of {} frames (fp32 torch-CPU restatement incl. v-trace)".format(steps, f),
            "ms_per_sgd_step": 1e3 * t / steps}


# ------------------------------------------------------------------------------------------------ GPU measurements
def device_report(busy):
    """Diagnostic only (never fails the line): what the box is -- host CPU, and the GPU's clocks / power as rocm-smi sees them
    WHILE `busy()` keeps the GPU running the headline loop (an idle MI355X reports its 107 MHz sleep clock).  Boxes of the
    same pool have measured 7.0 and 11 ms for the same update; this is what tells them apart in the record."""
    rep = {}
    pr = None
    try:
        cpu = subprocess.run("lscpu", shell=True, capture_output=True, text=True, timeout=5).stdout
        for line in cpu.splitlines():
            if line.startswith("Model name:"):
                rep["host_cpu"] = line.split(":", 1)[1].strip()
        pr = subprocess.Popen(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower", "--showperflevel",
                               "--showcomputepartition", "--showmemorypartition", "--json"],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        t0 = time.perf_counter()
        while pr.poll() is None and time.perf_counter() - t0 < 6.0:
            busy()
        txt = pr.communicate(timeout=5)[0]
        js = json.loads(txt[txt.index("{"):])
        card = js[sorted(js)[0]]
        for k, v in card.items():
            kk = k.strip(" :").lower()
            if "clock speed" in kk:
                rep[kk.split()[0] + "_mhz_busy"] = float("".join(ch for ch in v if ch.isdigit() or ch == "."))
            elif kk.startswith("max graphics package power"):
                rep["power_cap_w"] = float(v)
            elif kk.startswith("current socket graphics package power"):
                rep["power_w_busy"] = float(v)
            elif kk == "performance level":
                rep["perf_level"] = v
            elif "partition" in kk:
                rep[kk.replace(" ", "_")] = v
    except Exception as e:  # noqa: BLE001
        rep["note"] = "rocm-smi not usable here: %r" % (e,)
        if pr is not None and pr.poll() is None:
            pr.kill()
    return rep


def layer_rooflines(net, spec, rows, obs, idx, x6=True):
    """Every layer kernel of one SGD step timed live with HIP events on the launch stream (xt_net_time_layer);
    algorithmic FLOPs = 2*M*N*K per GEMM.  The arithmetic kind a launch is priced against is
    what the library reports for it (xt_last_launch_arith: fp32 MFMA, bf16x3 first-layer kernels, bf16x6, or a fused
    backward launch with an fp32 weight-gradient half and a bf16x6 input-gradient half).
    Returns {label: (ms, flops, arithmetic kind)}."""
    from xingtian_amd import lib as L
    handle = L.load()
    names = {0: "fp32", 1: "bf16x3", 2: "bf16x6", 3: "fp32+bf16x6"}
    kern = {}
    for li, lay in enumerate(spec.layers):
        mnk2 = 2.0 * rows * lay.OH * lay.OW * lay.N * lay.K
        ms = net.time_layer(li, 0, obs, idx, rows, 50)
        kern["L%d %s fwd" % (li, lay.name)] = (ms, mnk2, names[int(handle.xt_last_launch_arith())])
        if li == 0:
            ms = net.time_layer(0, 1, obs, idx, rows, 50)
            kern["L0 %s wgrad" % lay.name] = (ms, mnk2, names[int(handle.xt_last_launch_arith())])
        else:
            ms = net.time_layer(li, 3, obs, idx, rows, 50)
            kern["L%d %s dgrad+wgrad" % (li, lay.name)] = (ms, 2 * mnk2, names[int(handle.xt_last_launch_arith())])
    return kern


# peak of a fused backward launch = harmonic mean of its two halves' peaks (equal FLOPs in each half)
PEAK = {"fp32": FP32_MFMA_PEAK_TFLOPS, "bf16x3": BF16_MFMA_PEAK_TFLOPS / 3.0, "bf16x6": BF16_MFMA_PEAK_TFLOPS / 6.0,
        "fp32+bf16x6": 2.0 / (1.0 / FP32_MFMA_PEAK_TFLOPS + 6.0 / BF16_MFMA_PEAK_TFLOPS)}


# kernel symbol (substring) behind every roofline label, per workload: joins the live timings with the committed
# rocprofv3 --pmc summaries (profiles/r02_pmc_<workload>.json, tools/profile_round.sh)
KERNEL_OF = {
    "ppo": {"L0 shared_conv_layer_0 fwd": "conv_u8c4k8_fwd_flat_kernel", "L0 shared_conv_layer_0 wgrad": "conv_u8c4k8_wgrad_flat_kernel",
            "L1 shared_conv_layer_1 fwd": "igemm_fwd_kernel<128, 32, 4, 1, false, false, ",
            "L1 shared_conv_layer_1 dgrad+wgrad": "igemm_bwd_layer_kernel<128, 32, 4, 1, false, 128, 32, 4, 1, 2, 0",
            "L2 shared_conv_layer_2 fwd": "igemm_fwd_kernel<64, 64, 2, 2, false, false, 2",
            "L2 shared_conv_layer_2 dgrad+wgrad": "igemm_bwd_layer_kernel<64, 64, 2, 2, false, 128, 32, 4, 1, 0, 2",
            "L3 shared_hidden_mlp_0 fwd": "igemm_fwd_kernel<64, 64, 2, 2, false, false, 2",
            "L3 shared_hidden_mlp_0 dgrad+wgrad": "igemm_bwd_layer_kernel<64, 64, 2, 2, false, 64, 64, 2, 2, 0, 0"},
    "impala": {"L0 explore_agent/conv2d fwd": "conv_u8c4_same_fwd_kernel",
               "L0 explore_agent/conv2d wgrad": "conv_u8c4_same_wgrad_kernel",
               "L1 explore_agent/conv2d_1 fwd": "direct_fwd_kernel<1, 1, true, 512",
               "L1 explore_agent/conv2d_1 dgrad+wgrad": "igemm_bwd_layer_kernel<128, 32, 4, 1, true, 128, 32, 4, 1, 0, ",
               "L2 explore_agent/conv2d_2 fwd": "igemm_fwd_kernel<64, 64, 2, 2, false, false",
               "L2 explore_agent/conv2d_2 dgrad+wgrad": "igemm_bwd_layer_kernel<64, 64, 2, 2, false, 64, 64, 2, 2, 0, 0"},
}


PROFILE_ROUND = "r06"


def library_identity():
    """What binary the numbers of this line were measured on: the digest embedded in the LOADED library at build time
    (xt_build_sources_sha) next to the digest of the source tree beside it -- they differ when a stale prebuilt .so
    travelled with newer sources."""
    from xingtian_amd import lib as L
    built, tree = L.built_sources_sha(), L.kernel_sources_sha()
    return {"built_from_sources_sha": built, "source_tree_sha": tree, "stale_binary": built != tree, "abi": L.ABI_VERSION}


def pmc_row(workload, sym):
    """Counters of kernel `sym` from the committed rocprofv3 --pmc summary of this round
    (profiles/r04_pmc_<workload>.json, tools/profile_round.sh) -- ONLY if that summary was measured on the kernel sources
    the LOADED library was compiled from (the digest embedded in the binary, not the tree's): a stale profile, or a stale
    binary, is reported as such instead of being joined to fresh timings."""
    from xingtian_amd.lib import built_sources_sha
    rel = os.path.join("profiles", "{}_pmc_{}.json".format(PROFILE_ROUND, workload))
    path = os.path.join(ROOT, rel)
    if not (sym and os.path.exists(path)):
        return None, "no committed counter summary ({})".format(rel)
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        return None, "unreadable " + rel
    cur = built_sources_sha()
    if doc.get("kernel_sources_sha") != cur:
        return None, "stale: {} was measured on kernel sources {} (loaded library: {})".format(rel, doc.get("kernel_sources_sha"), cur)
    for row in doc.get("kernels", []):
        if sym in row["kernel"]:
            return row, "{}@{}".format(rel, cur)
    return None, "kernel not in " + rel


def box_health(roof):
    """Slow-box detector: on a healthy box a kernel's duration inside the replayed graph is at most its isolated
    launch-to-launch period (which contains a boundary); on the degraded boxes of this pool (one gpurun call in eight in
    round 3: 627 W instead of ~830 W) every kernel ran ~1.5x longer INSIDE the graph while the isolated timings stayed
    normal.  Flags the line when the median in-graph / isolated ratio exceeds 1.25."""
    ig, iso = roof.get("kernels_us_in_graph") or {}, roof.get("kernels_us_isolated") or {}
    ratios = sorted(ig[k] / iso[k] for k in ig if k in iso and iso[k] > 0)
    if not ratios:
        return {"degraded_box": None, "note": "no in-graph kernel averages in this run"}
    med = ratios[len(ratios) // 2]
    return {"degraded_box": bool(med > 1.25), "median_in_graph_over_isolated": round(med, 3),
            "max_in_graph_over_isolated": round(ratios[-1], 3)}


def roofline_of(kern, workload="ppo", in_graph=None):
    """Dominant layer kernel of one SGD step.  ``kern``: ISOLATED timings (xt_net_time_layer: back-to-back launches of
    one kernel, HIP events on the launch stream, live in this run) -- the basis of achieved / frac.  ``in_graph``:
    {symbol substring: us} averages of the same kernels inside the replayed hipGraph of this run (rocprofv3
    --kernel-trace --stats of a short re-run), when they could be collected."""
    dom = max(kern, key=lambda k: kern[k][0])
    ms, flops, kind = kern[dom]
    ach = flops / (ms * 1e-3) / 1e12
    table = KERNEL_OF["ppo" if workload == "ppo" else "impala"]
    sym = table.get(dom)
    row, src = pmc_row(workload, sym)
    traffic = row["hbm_side_MB"] * 1048576.0 if row and row.get("hbm_side_MB") is not None else None
    out = {"bound": "mfma", "kernel": dom, "kernel_symbol": sym, "arith": kind, "achieved": ach, "peak": PEAK[kind],
           "unit": "TFLOP/s", "frac": ach / PEAK[kind], "traffic": traffic,
           "mfma_pipe_util_pmc": row.get("mfma_pipe_util") if row else None, "pmc_source": src,
           "flop_per_launch": flops, "avg_launch_ms": ms, "timing": "isolated (xt_net_time_layer, HIP events, this run)",
           "kernels_us_isolated": {k: round(v[0] * 1e3, 2) for k, v in kern.items()},
           "sum_layer_kernels_us": round(sum(v[0] for v in kern.values()) * 1e3, 1)}
    if in_graph:
        ig = {}
        for label, sub in table.items():
            hit = [us for name, us in in_graph.items() if sub in name]
            if hit and label in kern:
                ig[label] = round(sum(hit) / len(hit), 2)
        out["kernels_us_in_graph"] = ig
        if ig:
            # The kernel's own duration INSIDE the update (begin -> end timestamps of rocprofv3's kernel trace, live in this
            # run) is what `achieved` / `frac` are priced with when it is available: the isolated figure is the period of
            # back-to-back launches of one kernel, i.e. it also contains a launch boundary (~2.5 us) that belongs to no kernel,
            # and it is what the committed profiles/r03_kernel_stats_*.csv must agree with.  Both views stay in the line.
            dom_ig = max((k for k in ig), key=lambda k: ig[k])
            f_ig, kind_ig = kern[dom_ig][1], kern[dom_ig][2]
            out["isolated"] = {"kernel": dom, "avg_launch_us": ms * 1e3, "achieved": ach, "frac": ach / PEAK[kind],
                               "note": "period of 50 back-to-back launches (HIP events): kernel + one launch boundary"}
            ach_ig = f_ig / (ig[dom_ig] * 1e-6) / 1e12
            row_ig, src_ig = pmc_row(workload, table.get(dom_ig))
            out.update({"kernel": dom_ig, "kernel_symbol": table.get(dom_ig), "arith": kind_ig, "achieved": ach_ig,
                        "peak": PEAK[kind_ig], "frac": ach_ig / PEAK[kind_ig], "flop_per_launch": f_ig,
                        "avg_launch_ms": ig[dom_ig] * 1e-3,
                        "traffic": row_ig["hbm_side_MB"] * 1048576.0 if row_ig and row_ig.get("hbm_side_MB") is not None else None,
                        "mfma_pipe_util_pmc": row_ig.get("mfma_pipe_util") if row_ig else None, "pmc_source": src_ig,
                        "timing": "in-graph: the kernel's average duration inside the replayed hipGraph of this run "
                                  "(rocprofv3 --kernel-trace --stats of a short re-run, see in_graph_source)"})
    return out


def in_graph_kernel_stats(workload, timeout=240):
    """rocprofv3 --kernel-trace --stats of a short --quick re-run of this script (the same hipGraph replay): per-kernel
    average durations INSIDE the graph.  Returns ({kernel name: us}, note)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="xt_ig_")
    cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.abspath(__file__),
           "--workload", workload, "--steps", "6", "--warmup", "2", "--quick", "--no-cpu-baseline", "--no-in-graph-stats"]
    try:
        subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       timeout=timeout, check=True)
        # the raw trace first: mean WITHOUT the launches longer than 10 x the median (one stalled launch in 211 once turned a
        # 19.4 us kernel into a "113.9 us" one, profiles/r05_kernel_stats_pong_impala_speedup.csv); rocprofv3's own stats
        # table (mean / min / max only) is the fall-back
        per = {}
        for path in glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if "xt::" in r["Kernel_Name"]:
                    per.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        res, outliers = {}, 0
        for name, v in per.items():
            if len(v) >= 20:
                a = np.asarray(v, np.float64)
                keep = a <= 10.0 * np.median(a)
                outliers += int((~keep).sum())
                res[name] = float(a[keep].mean()) / 1e3
        note = "rocprofv3 --kernel-trace of `bench.py --workload {} --quick --steps 6` (this box, this build): mean of the launches " \
               "within 10 x the kernel's median ({} outlier launch(es) dropped)".format(workload, outliers)
        if not res:
            files = glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True)
            if not files:
                return None, "rocprofv3 wrote neither a kernel trace nor a kernel_stats.csv"
            for r in csv.DictReader(open(files[0])):
                if "xt::" in r["Name"] and int(r["Calls"]) >= 20:
                    res[r["Name"]] = float(r["AverageNs"]) / 1e3
            note = "rocprofv3 --kernel-trace --stats of `bench.py --workload {} --quick --steps 6` (this box, this build)".format(workload)
        return res, note
    except (subprocess.SubprocessError, OSError) as exc:
        return None, "rocprofv3 run failed: {!r}".format(exc)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def bench_e2e_ppo(env_num, min_seconds=1.0, max_updates=40, via_ring=False, learner_gae=False, handover="get_weights", warm=3):
    """(rollout samples consumed by one Algorithm.train()) / (wall time of prepare_data x env_num +
    train() incl. the H2D of the uint8 rollout + get_weights D2H), through the plugin classes exactly as
    xt/framework/learner.py:306-313,346-348,361-363 drives them.  Host arrays are plain (pageable) numpy."""
    from xingtian_amd.algorithm import alg_builder
    from xingtian_amd import ops      # the actors' GAE for the synthetic trajectories (not timed): the product's own bit-exact kernel
    model_info = {"actor": {"model_name": "PpoCnn", "state_dim": list(STATE_DIM), "action_dim": A_DIM,
                            "input_dtype": "uint8",
                            "model_config": dict(CFG, SUMMARY=False, VF_SHARE_LAYERS=True, activation="relu",
                                                 hidden_sizes=list(HIDDEN), action_type="Categorical", SEED=0)}}
    alg = alg_builder("PPO", model_info, {"instance_num": env_num, "agent_num": 1})
    obs, action, logp, value, reward, done = synth_rollout(100 + env_num, env_num)
    trajs = []
    adv_all, oldv_all, tgt_all = ops.gae(value, reward, done, 0.99, 0.95)      # [env_num, T] float64 / float32 / float64
    for i in range(env_num):
        a, ov, tg = adv_all[i].reshape(-1, 1), oldv_all[i].reshape(-1, 1), tgt_all[i].reshape(-1, 1)
        sl = slice(i * T_LEN, (i + 1) * T_LEN)
        if learner_gae:
            # the trajectory as the explorer holds it BEFORE data_proc: the learner stages value / reward / done with the
            # frames and runs ONE xt_gae_f64_ragged over the rollout on the device inside train() (the product's GAE path)
            trajs.append({"cur_state": obs[sl], "action": action[sl], "logp": logp[sl].reshape(-1, 1),
                          "value": value[i].reshape(-1, 1), "reward": list(reward[i]), "done": list(done[i])})
        else:
            trajs.append({"cur_state": obs[sl], "action": action[sl], "logp": logp[sl].reshape(-1, 1), "adv": a,
                          "old_value": ov, "target_value": tg})
    t_prep = t_train = t_w = 0.0
    updates = 0
    ring, wire, wring = None, None, None
    if handover == "publish":
        from xingtian_amd import transport
        wring = transport.WeightsRing(slot_bytes=8 << 20, slots=3)
        if not wring.pin():
            wring.close()
            return {"skipped": "hipHostRegister of the weights ring failed"}
    if via_ring:
        # the trajectories arrive as encoded messages in a PINNED shared-memory ring (xingtian_amd/transport.py): what the
        # learner process sees when explorers feed it; the explorer-side copy INTO the ring is not learner time
        from xingtian_amd import transport
        ring = transport.ShmRing(slots=8, slot_bytes=4 << 20)
        if not ring.pin():
            ring.close()
            return {"skipped": "hipHostRegister of the shared-memory ring failed"}
        wire = [bytes(transport.encode({"cmd": "train", "explorer_id": i}, dict(tr, reward=[0.0] * T_LEN, done=[False] * T_LEN)))
                for i, tr in enumerate(trajs)]

    def feed():
        """-> learner-side seconds spent in prepare_data for the whole rollout"""
        if ring is None:
            t0 = time.perf_counter()
            for tr in trajs:
                alg.prepare_data(tr)
            return time.perf_counter() - t0
        spent = 0.0
        for m in wire:
            if not ring.send_bytes(m, block=False):     # explorer side; a full ring = copies still reading the slots
                t0 = time.perf_counter()
                ring.drain()                            # (learner time: it waits for its own DMAs)
                spent += time.perf_counter() - t0
                assert ring.send_bytes(m, block=False)
            t0 = time.perf_counter()
            ring.recv_into(alg.prepare_data)            # learner side: decode -> DMA out of the pinned slot, no wait
            spent += time.perf_counter() - t0
        return spent

    def one(timed):
        nonlocal t_prep, t_train, t_w, updates
        dt_prep = feed()
        t1 = time.perf_counter()
        t0 = t1 - dt_prep
        loss = alg.train(episode_num=updates)
        t2 = time.perf_counter()
        if wring is not None:
            assert alg.publish_weights(wring) > 0
        else:
            w = alg.get_weights()
            assert len(w) == 12
        t3 = time.perf_counter()
        assert np.isfinite(loss)
        if timed:
            t_prep += t1 - t0; t_train += t2 - t1; t_w += t3 - t2; updates += 1

    for _ in range(warm):
        one(False)
    t_begin = time.perf_counter()
    while updates < max_updates and (updates < 5 or time.perf_counter() - t_begin < min_seconds):
        one(True)
    total = t_prep + t_train + t_w
    n = env_num * T_LEN
    bytes_h2d = n * (int(np.prod(STATE_DIM)) + 4 + 4 + 8 + 4 + 8)
    if ring is not None:
        ring.close()
    if wring is not None:
        wring.close()
    return {"env_num": env_num, "gae": "learner GPU (one xt_gae_f64_ragged per rollout, inside train())" if learner_gae
            else "actor side (adv / old_value / target_value arrive with the trajectory, the reference's protocol)",
            "weights_handover": "publish_weights(pinned WeightsRing): one D2H into the slot" if wring is not None
            else "get_weights(): dict of private arrays (the reference API)", "env_steps_per_update": n, "updates": updates,
            "value": FRAME_SKIP * n * updates / total, "unit": "env-frames/s", "ms_per_update": 1e3 * total / updates,
            "prepare_data_ms": 1e3 * t_prep / updates, "train_ms": 1e3 * t_train / updates,
            "get_weights_ms": 1e3 * t_w / updates, "h2d_bytes_per_update": bytes_h2d,
            "h2d_ms_at_55GBps": 1e3 * bytes_h2d / 55e9, "stream_ingest": bool(alg.actor.stream_ingest),
            "path": ("alg_builder('PPO') -> ShmRing.recv_into(prepare_data) x {} (encoded messages in a hipHostRegister'ed "
                     "shared-memory ring -> DMA straight to HBM, no learner-side host copy) -> train() -> get_weights()"
                     if via_ring else
                     "alg_builder('PPO') -> prepare_data x {} (pageable numpy -> pinned staging -> async H2D) -> "
                     "train() -> get_weights() (one pinned D2H)").format(env_num)}


def bench_impala(key, steps, warmup, with_cpu, in_graph=False, quick=False):
    """HBM-resident and plugin-path measurements of one IMPALA configuration (secondary workloads)."""
    from xingtian_amd.algorithm import alg_builder
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    w = IMPALA[key]
    dev = torch.device("cuda", torch.cuda.current_device())
    f, trains = w["frames_per_train"], w["trains"]
    n = f * trains
    data = synth_impala(7, n, w["dim"], w["a_dim"])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dobs, dbp, dact = d(data["obs"]), d(data["logit"]), d(data["action"])
    ddone, drew = d(data["done"].astype(np.uint8)), d(data["reward"].astype(np.float32))
    spec = netspec.impala_cnn_opt((w["dim"], w["dim"], 4), w["a_dim"], w["mean"], w["std"], "uint8")
    net = HipActorCritic(spec, max_batch=f, seed=0)
    cfg = net.make_impala_cfg(w["lr"], 40.0, w["t_len"])

    def one_step():      # `trains` learner trains (one BATCH_SIZE chunk each) enqueued by ONE C call, hipGraph replay
        net.impala_train(cfg, dobs, f, dbp, dact, ddone, drew, use_graph=True)

    for _ in range(warmup):
        one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert torch.isfinite(net.params).all()
    us_per_train = 1e6 * el / (steps * trains)
    kern = layer_rooflines(net, spec, f, dobs, None, x6=True)
    # in-graph per-kernel averages of THIS workload (one short rocprofv3 --kernel-trace re-run), so that its roofline is
    # priced with the kernel's duration inside the update like the headline's -- the isolated launch-to-launch period of
    # xt_net_time_layer charges a split-K forward for its finish launch and every kernel for a boundary
    ig, ig_note = in_graph_kernel_stats(key) if in_graph else (None, "skipped")
    roof = roofline_of(kern, "breakout_impala" if key.startswith("breakout") else "pong_impala_speedup", ig)
    roof["in_graph_source"] = ig_note
    out = {"workload": w["name"], "metric": "learner env-frames/sec", "unit": "env-frames/s", "dtype": "fp32",
           "value": FRAME_SKIP * n * steps / el, "us_per_train": us_per_train, "frames_per_train": f,
           "trains_per_step": trains, "steps": steps,
           "update_tflops": w["mflop"] * 1e6 * f / (us_per_train * 1e-6) / 1e12,
           "update_frac_of_fp32_mfma_peak": w["mflop"] * 1e6 * f / (us_per_train * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
           "hip_graph": True, "roofline": roof, "semantic_change": bool(w.get("semantic_change", False))}
    del net
    if quick:
        return out
    # ---- plugin path, as xt/framework/learner.py:306-366 drives it: prepare_data x k -> train() -> every
    # train_per_checkpoint-th train the weights go out.  Two hand-overs are measured: `get_weights()` (the reference API:
    # a dict of PRIVATE arrays, then whatever the caller does with it) and `publish_weights(ring)` (the fan-out itself:
    # one D2H straight into a page-locked transport.WeightsRing slot, no host copy on the learner)
    msgs_per_train = w.get("msgs_per_train", 1 if key == "breakout_impala" else 4)
    tpc = w.get("train_per_checkpoint", 1)
    fm = f // msgs_per_train
    msgs = []
    for i in range(min(trains, 8) * msgs_per_train):
        sl = slice(i * fm, (i + 1) * fm)
        msgs.append({"cur_state": data["obs"][sl], "logit": data["logit"][sl], "action": data["action"][sl],
                     "done": list(data["done"][sl]), "reward": list(data["reward"][sl])})

    def plugin_run(handover, async_loss=False, lag=0):
        from xingtian_amd import transport
        model_info = {"actor": {"model_name": "ImpalaCnnOpt", "state_dim": [w["dim"], w["dim"], 4], "input_dtype": "uint8",
                                "state_mean": w["mean"], "state_std": w["std"], "action_dim": w["a_dim"],
                                "model_config": {"LR": w["lr"], "sample_batch_step": w["t_len"], "grad_norm_clip": 40.0,
                                                 "SEED": 0, "ASYNC_LOSS": bool(async_loss)}}}
        alg = alg_builder("IMPALAOpt", model_info, {"instance_num": 32, "agent_num": 1,
                                                   "prepare_times_per_train": msgs_per_train, "train_per_checkpoint": tpc,
                                                   "BATCH_SIZE": max(f, 512) if key.startswith("breakout_impala") else f})
        ring = None
        if handover == "publish":
            ring = transport.WeightsRing(slot_bytes=8 << 20, slots=3)
            if not ring.pin():
                ring.close()
                return {"skipped": "hipHostRegister of the weights ring failed"}
        t_prep = t_train = t_w = 0.0
        cnt = 0

        def one_train(i, timed):
            nonlocal t_prep, t_train, t_w, cnt
            t0 = time.perf_counter()
            for k in range(msgs_per_train):
                alg.prepare_data(msgs[(i * msgs_per_train + k) % len(msgs)])
            t1 = time.perf_counter()
            loss = alg.train(episode_num=i)
            t2 = time.perf_counter()
            if alg.checkpoint_ready(i):         # train_count BEFORE its increment, as learner.py:361 passes it
                if ring is not None:
                    assert alg.publish_weights(ring, lag=lag) > 0
                else:
                    wts = alg.get_weights()
                    assert len(wts) >= 8
            t3 = time.perf_counter()
            assert np.isfinite(loss)
            if timed:
                t_prep += t1 - t0; t_train += t2 - t1; t_w += t3 - t2; cnt += 1

        for i in range(6):
            one_train(i, False)
        tb = time.perf_counter()
        i = 0
        while cnt < 400 and (cnt < 20 or time.perf_counter() - tb < 1.0):
            one_train(i, True)
            i += 1
        torch.cuda.synchronize()
        if ring is not None:
            ring.close()
        tot = t_prep + t_train + t_w
        return {"value": FRAME_SKIP * f * cnt / tot, "unit": "env-frames/s", "trains": cnt,
                "ms_per_train": 1e3 * tot / cnt, "prepare_data_ms": 1e3 * t_prep / cnt, "train_ms": 1e3 * t_train / cnt,
                "weights_ms": 1e3 * t_w / cnt, "train_per_checkpoint": tpc,
                "async_loss": bool(async_loss), "weights_lag_trains": int(lag) * int(tpc),
                "path": "alg_builder('IMPALAOpt') -> prepare_data x {} -> train() -> every {} train(s): {}".format(
                    msgs_per_train, tpc, "publish_weights(pinned WeightsRing): one D2H into the slot" if handover == "publish"
                    else "get_weights(): dict of private arrays")}

    out["e2e"] = plugin_run("get_weights")
    out["e2e_publish"] = plugin_run("publish")
    # model_config ASYNC_LOSS: train() returns the PREVIOUS train's loss instead of waiting for its own, so the next
    # message is staged / copied while the GPU runs this train (flagged: the logged loss lags one train; weights do not)
    out["e2e_publish_async_loss"] = dict(plugin_run("publish", async_loss=True),
                                         note="model_config ASYNC_LOSS: the reported loss lags one train; weights handed out are current")
    # fully pipelined learner loop (FLAGGED deviation): ASYNC_LOSS + publish_weights(lag=1) -- the weights handed out after
    # train k are those of train k-1 (their copy landed long ago), nothing in the loop waits for the GPU; what an
    # asynchronous algorithm like IMPALA tolerates by design (v-trace corrects the policy lag), not what learner.py does
    out["e2e_pipelined"] = dict(plugin_run("publish", async_loss=True, lag=1), semantic_change=True,
                                note="ASYNC_LOSS + publish_weights(lag=1): the loss lags one train, the published weights one publish "
                                     "interval (= train_per_checkpoint trains)")
    # ---- (round 6) the learner fed as the framework feeds it: a producer process -> pinned shared-memory ring ->
    # transport.Prefetcher (every message staged to HBM when it ARRIVES: train k+1's H2D under the GPU's train k) -> the
    # reference's loop; weights: the D2H the update enqueued into a pinned WeightsRing slot, committed by the ring's helper
    # thread when it lands.  Same messages, same order, same trains, same weights -- no flag.  The blocking form of the same
    # ring-fed loop sits next to it.
    out["e2e_ring_prefetch"] = dict(impala_ring_loop(w, fm, msgs_per_train, tpc, n_prod=2, seconds=1.0),
                                    path="producer processes -> pinned RingSet -> transport.Prefetcher (inline: the learner thread stages the "
                                         "next message itself while the device trains) -> prepare_data x {} -> train() "
                                         "-> every {} train(s): publish_weights(pinned WeightsRing, committer thread)".format(msgs_per_train, tpc))
    out["e2e_ring_blocking"] = dict(impala_ring_loop(w, fm, msgs_per_train, tpc, n_prod=2, seconds=0.7, prefetch=False,
                                                     async_commit=False),
                                    path="the same rings, learner thread receives + waits for the weights D2H itself")
    out["e2e_ring_prefetch_thread"] = dict(impala_ring_loop(w, fm, msgs_per_train, tpc, n_prod=2, seconds=0.7, inline=False),
                                           path="as e2e_ring_prefetch with the staging on a THREAD of its own (transport.Prefetcher("
                                                "inline=False)): two Python threads on one interpreter lock")
    out["e2e_ring_prefetch_python_lists"] = dict(impala_ring_loop(w, fm, msgs_per_train, tpc, n_prod=2, seconds=0.7, pack_lists=False),
                                                 path="as e2e_ring_prefetch, done / reward on the wire as python lists (msgpack): "
                                                      "the sender did not use transport.encode(pack_lists=True)")
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline_impala(w, data)
    return out


def bench_impala_dp(key, rank, world, dev, dist, trains=40, warmup=5):
    """IMPALA data parallel (BASELINE configs[4] direction): the sum-form loss -> gradients are SUMMED, no scaling.
    strict: the chunk's trajectories are split into whole-trajectory shards (same chunk on every rank); weak: every
    rank trains on its own full chunk (global chunk = world x BATCH_SIZE frames: flagged semantic change).  A chunk of
    one trajectory (breakout_impala) cannot be sharded: weak only."""
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    from xingtian_amd.parallel import allreduce_sum_, dp_impala_step
    w = IMPALA[key]
    f, t_len = w["frames_per_train"], w["t_len"]
    ntraj = f // t_len
    spec = netspec.impala_cnn_opt((w["dim"], w["dim"], 4), w["a_dim"], w["mean"], w["std"], "uint8")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    res = {"workload": w["name"], "unit": "env-frames/s", "n_gpus": world}
    for mode in (("strict", "weak") if ntraj >= world else ("weak",)):
        data = synth_impala(7 if mode == "strict" else 70 + rank, f, w["dim"], w["a_dim"])
        obs, bp, act = d(data["obs"]), d(data["logit"]), d(data["action"])
        done, rew = d(data["done"].astype(np.uint8)), d(data["reward"].astype(np.float32))
        net = HipActorCritic(spec, max_batch=f, device=str(dev), seed=0)
        cfg = net.make_impala_cfg(w["lr"], 40.0, t_len)

        def one():
            if mode == "strict":
                dp_impala_step(net, cfg, w["lr"], 40.0, obs, bp, act, done, rew, ntraj, t_len, rank, world)
            else:
                net.impala_step(cfg, obs, bp, act, done, rew, apply=False)
                allreduce_sum_(net.grads)
                net.apply(w["lr"], 40.0, grad_scale=1.0)

        for _ in range(warmup):
            one()
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(trains):
            one()
        dist.barrier(); torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        frames = f * (world if mode == "weak" else 1) * trains
        res[mode] = {"value": FRAME_SKIP * frames / float(el.item()), "us_per_train": 1e6 * float(el.item()) / trains,
                     "frames_per_train_global": f * (world if mode == "weak" else 1),
                     "trajectories_per_rank": ntraj if mode == "weak" else "{}..{}".format(ntraj // world, -(-ntraj // world))}
        assert torch.isfinite(net.params).all()
        del net
    return res


def bench_env_num_256(spec, dev, updates=3):
    """BASELINE.json configs[3] at its stated per-update scale on ONE GPU: Breakout PPO with env_num = 256
    -> 32 768 samples = 925 MB of uint8 frames per update, 4 x ceil(32768 / 320) = 412 SGD steps
    (xt/model/ppo/ppo.py:111-132).  HBM-resident (GAE + xt_net_ppo_train, hipGraph replay) and through the plugin classes
    (prepare_data x 256 -> train() -> weights)."""
    from xingtian_amd import lib as L
    from xingtian_amd.model.hip_net import HipActorCritic
    lib = L.load()
    env_num = 256
    obs, action, logp, value, reward, done = synth_rollout(256, env_num)
    n = obs.shape[0]
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    net = HipActorCritic(spec, max_batch=CFG["BATCH_SIZE"], device=str(dev), seed=0)
    cfg = net.make_ppo_cfg(CFG)
    d_obs, d_act, d_logp = d(obs), d(action), d(logp)
    d_value, d_reward, d_done = d(value), d(reward), d(done.astype(np.uint8))
    d_adv = torch.empty((n,), dtype=torch.float64, device=dev)
    d_tgt = torch.empty((n,), dtype=torch.float64, device=dev)
    d_oldv = torch.empty((n,), dtype=torch.float32, device=dev)
    d_perm = torch.empty((CFG["NUM_SGD_ITER"], n), dtype=torch.int32, device=dev)
    pin = torch.empty((CFG["NUM_SGD_ITER"], n), dtype=torch.int32, pin_memory=True)
    rng = np.random.default_rng(77)

    def one_update():
        torch.cuda.current_stream().synchronize()          # (one pinned block: its previous H2D has finished)
        inds = np.arange(n)
        p = pin.numpy()
        for ep in range(CFG["NUM_SGD_ITER"]):
            rng.shuffle(inds)
            p[ep] = inds
        d_perm.copy_(pin, non_blocking=True)
        L.check(lib.xt_gae_f64(L.ptr(d_value), L.ptr(d_reward), L.ptr(d_done), L.ptr(d_adv), L.ptr(d_tgt), L.ptr(d_oldv),
                               env_num, T_LEN, 0.99, 0.95, L.stream_ptr()), "gae")
        net.ppo_train(cfg, d_obs, d_perm, d_act, d_logp, d_adv, d_oldv, d_tgt, use_graph=True)

    one_update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(updates):
        one_update()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / updates
    assert torch.isfinite(net.params).all()
    sgd = CFG["NUM_SGD_ITER"] * ((n + CFG["BATCH_SIZE"] - 1) // CFG["BATCH_SIZE"])
    out = {"workload": "BASELINE configs[3] per-update scale on 1 GPU: Breakout PPO env_num=256, 32768 samples (925 MB uint8), "
                       "{} SGD steps of B=320".format(sgd),
           "value": FRAME_SKIP * n / el, "unit": "env-frames/s", "ms_per_update": 1e3 * el, "sgd_steps": sgd,
           "us_per_sgd_step": 1e6 * el / sgd, "rollout_bytes": int(obs.nbytes)}
    del net, d_obs
    torch.cuda.empty_cache()
    for key, kw in (("e2e", dict()), ("e2e_pinned_ring", dict(via_ring=True, handover="publish"))):
        try:
            log("env_num 256:", key)
            r = bench_e2e_ppo(env_num, min_seconds=0.2, max_updates=5, warm=2, **kw)
        except Exception as exc:      # noqa: BLE001 -- a diagnostic block must not take the line with it
            r = {"error": repr(exc)[:200]}
        out[key] = r
    best = out["e2e_pinned_ring"] if "value" in out.get("e2e_pinned_ring", {}) else out.get("e2e", {})
    if "value" in best:
        out.update({"value_e2e": best["value"], "e2e_ms_per_update": best["ms_per_update"],
                    "e2e_prepare_data_ms": best["prepare_data_ms"], "e2e_train_ms": best["train_ms"],
                    "h2d_share_of_e2e": best["h2d_ms_at_55GBps"] / best["ms_per_update"]})
    return out


def impala_ring_loop(w, fm, msgs_per_train, tpc, n_prod, seconds, prefetch=True, async_commit=True, pinned=True, slots=4,
                     min_trains=20, gate=True, model_config=None, pack_lists=True, strict=False, inline=None):
    """The IMPALAOpt plugin pair fed as a learner is fed (xt/framework/learner.py:298-380): `n_prod` producer PROCESSES push
    pre-encoded rollout messages of `fm` frames into their own shared-memory ring (transport.RingSet); the learner loop is
    the reference's -- recv + prepare_data x msgs_per_train -> train() -> every tpc-th train the weights go out.
    prefetch: a transport.Prefetcher thread owns the rings and stages every message to HBM as it arrives (the H2D of train
    k+1 under the GPU's train k); async_commit: the D2H of the new weights, enqueued by the update into a pinned
    WeightsRing slot, is made visible by the ring's committer thread when it lands (the learner does not sit through it; the
    weights are THIS train's).  Neither changes what is trained or published.  -> dict(messages_per_s, trains_per_s, ...)"""
    import multiprocessing as mp
    from xingtian_amd import transport
    from xingtian_amd.algorithm import alg_builder
    data = synth_impala(11, fm, w["dim"], w["a_dim"])
    # done / reward leave the agent as per-step python lists (xt/agent/impala); pack_lists: the SENDER ships them as typed arrays
    # (transport.encode(pack_lists=True), an explorer-side option of this transport) -- the learner's serial staging thread then
    # gets zero-copy views instead of 2 x T boxed scalars through msgpack + np.asarray per message
    wire = bytes(transport.encode({"cmd": "train"}, {"cur_state": data["obs"], "logit": data["logit"], "action": data["action"],
                                                     "done": list(data["done"]), "reward": list(data["reward"])},
                                  pack_lists=pack_lists))
    slot_bytes = (len(wire) + (1 << 16)) // 4096 * 4096
    ctx = mp.get_context("fork")
    try:
        st = os.statvfs("/dev/shm")
        need, free = n_prod * slots * slot_bytes * 1.1 + (64 << 20), st.f_bavail * st.f_frsize
    except OSError:
        need, free = 0, 1
    if free < need:                 # (a write beyond the tmpfs capacity is a SIGBUS, not an exception: check first)
        return {"skipped": "/dev/shm has {:.0f} MB free, {} rings of {} x {:.1f} MB need {:.0f} MB".format(
            free / 1e6, n_prod, slots, slot_bytes / 1e6, need / 1e6)}
    try:
        rs = transport.RingSet(n_prod, slots=slots, slot_bytes=slot_bytes)
    except OSError as exc:
        return {"skipped": repr(exc)[:120]}
    is_pinned = bool(rs.pin()) if (pinned and n_prod <= 128) else False    # (hipHostRegister of 512 segments: not worth the set-up)
    stop = ctx.Value("i", 0)
    backoff = min(0.005, max(0.0002, 0.25 * n_prod / 8000.0))      # ~1/4 of a producer's turn at ~8k messages/s
    procs = [ctx.Process(target=_scan_producer, args=(rs.names[i], slots, slot_bytes, wire, stop, backoff), daemon=True)
             for i in range(n_prod)]
    for p_ in procs:
        p_.start()
    model_info = {"actor": {"model_name": "ImpalaCnnOpt", "state_dim": [w["dim"], w["dim"], 4], "input_dtype": "uint8",
                            "state_mean": w["mean"], "state_std": w["std"], "action_dim": w["a_dim"],
                            "model_config": dict({"LR": w["lr"], "sample_batch_step": w["t_len"], "grad_norm_clip": 40.0,
                                                  "SEED": 0}, **(model_config or {}))}}
    alg = alg_builder("IMPALAOpt", model_info, {"instance_num": max(n_prod, 1), "agent_num": 1,
                                               "prepare_times_per_train": msgs_per_train, "train_per_checkpoint": tpc,
                                               "BATCH_SIZE": max(fm * msgs_per_train, 512) if w["dim"] == 84 else fm * msgs_per_train})
    wring = transport.WeightsRing(slot_bytes=8 << 20, slots=4)
    wpin = wring.pin()
    if wpin and async_commit:
        wring.start_committer()
        alg.actor.net.attach_weights_ring(wring)
    src = transport.Prefetcher(rs, alg, gate=gate, strict=strict, inline=inline) if prefetch else rs
    sink = lambda d_, ctr_info=None: alg.prepare_data(d_, ctr_info=ctr_info)   # noqa: E731
    trains = 0
    t_recv = t_train = t_w = 0.0
    out = {}
    try:
        def one_train(timed):
            nonlocal trains, t_recv, t_train, t_w
            t0 = time.perf_counter()
            got = src.recv_many_into(sink, msgs_per_train, timeout=20.0)
            if got != msgs_per_train:
                raise RuntimeError("producers delivered {} of {} messages in 20 s".format(got, msgs_per_train))
            t1 = time.perf_counter()
            loss = alg.train(episode_num=trains)
            t2 = time.perf_counter()
            if alg.checkpoint_ready(trains):
                alg.publish_weights(wring) if wpin else alg.get_weights()
            t3 = time.perf_counter()
            trains += 1
            if timed:
                t_recv += t1 - t0; t_train += t2 - t1; t_w += t3 - t2
            return loss
        for _ in range(8):
            one_train(False)
        trains = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds or trains < min_trains:
            loss = one_train(True)
        el = time.perf_counter() - t0
        assert np.isfinite(loss)
        if wpin:
            published = wring.drain()
            assert published >= 1
        out = {"messages_per_s": msgs_per_train * trains / el, "trains_per_s": trains / el,
               "value": FRAME_SKIP * fm * msgs_per_train * trains / el, "unit": "env-frames/s", "trains": trains,
               "ms_per_train": 1e3 * el / trains, "recv_prepare_ms": 1e3 * t_recv / trains, "train_ms": 1e3 * t_train / trains,
               "weights_ms": 1e3 * t_w / trains, "train_per_checkpoint": tpc, "producers": n_prod,
               "pinned_rings": is_pinned, "prefetch": bool(prefetch), "async_weights_commit": bool(wpin and async_commit),
               "lists_packed_by_sender": bool(pack_lists),
               "served_min_max": [int(min(rs.served)), int(max(rs.served))]}
    except Exception as exc:      # noqa: BLE001
        out = {"error": repr(exc)[:200]}
    finally:
        if prefetch:
            src.close()
        stop.value = 1
        deadline = time.perf_counter() + 5.0
        for p_ in procs:
            p_.join(max(0.0, deadline - time.perf_counter()))
        for p_ in [q for q in procs if q.is_alive()]:
            p_.terminate()
        torch.cuda.synchronize()
        alg.actor.net.attach_weights_ring(None)
        wring.close()
        rs.close()
        del alg
        torch.cuda.empty_cache()
    return out


def _scan_producer(name, slots, slot_bytes, wire, stop_flag, backoff):
    """explorer stand-in of bench_actor_scan: sends ONE pre-encoded rollout message over and over until told to stop"""
    from xingtian_amd import transport
    ring = transport.RingSet.attach(name, slots=slots, slot_bytes=slot_bytes)
    while not stop_flag.value:
        if not ring.send_bytes(wire, block=False):
            time.sleep(backoff)            # ring full: the learner is the bottleneck; do not burn the host's cores polling
    ring.close()


def bench_actor_scan(counts=(8, 32, 128, 512), seconds=1.0):
    """BASELINE.json configs[4] (examples/pong_impala_speedup.yaml:39 "512 actors -> throughput scan") on the learner side:
    P producer PROCESSES (one shared-memory ring each, transport.RingSet) push pre-encoded 250-frame rollout messages
    (5 envs x T=50, 42x42x4 uint8, A=6) as fast as the learner drains them; the learner is the plugin pair
    (IMPALAOpt: prepare_data x 4 -> train() on 1000 frames -> every 3rd train the weights go out through a pinned
    WeightsRing), fed through a transport.Prefetcher (round 6: every message goes to HBM when it arrives, the weights D2H is
    committed by the ring's helper thread).  Reports messages/s and env-frames/s per P and where it saturates, plus the
    round-5 loop (learner thread receives and waits itself) at P = 8.  The producers do no environment stepping: this scans the
    LEARNER's ingest + update capacity, which is what bounds a 512-actor run."""
    w = IMPALA["pong_impala_speedup"]
    fm, msgs_per_train, tpc = 250, 4, 3
    res = {"frames_per_message": fm, "scan": {}}
    cores = host_cores()
    for n_prod in counts:
        log("actor scan:", n_prod, "producers")
        res["scan"][str(n_prod)] = impala_ring_loop(w, fm, msgs_per_train, tpc, n_prod, seconds, slots=2)
        log("  ->", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in res["scan"][str(n_prod)].items()})
    res["blocking_loop_at_8"] = impala_ring_loop(w, fm, msgs_per_train, tpc, 8, seconds, prefetch=False, async_commit=False, slots=2)
    ok = {int(k): v["messages_per_s"] for k, v in res["scan"].items() if "messages_per_s" in v}
    if ok:
        peak_p = max(ok, key=lambda k: ok[k])
        res.update({"messages_per_s_peak": ok[peak_p], "producers_at_peak": peak_p, "host_cores": cores,
                    "value_at_512": res["scan"].get("512", {}).get("value"),
                    "messages_per_s_blocking_loop_at_8": res["blocking_loop_at_8"].get("messages_per_s"),
                    "saturates_at": min((k for k in ok if ok[k] >= 0.9 * ok[peak_p]), default=peak_p),
                    "note": "learner-side scan: producers replay one encoded message; saturation = the learner's "
                            "staging + H2D + update rate, not the actors"})
    return res


def run_section(name, timeout=240):
    """One block of the line in a FRESH process (`bench.py --section <name>`): the env_num 256 block page-locks ~2 GB of
    staging and the actor scan forks up to 512 producers -- forking from a process that holds gigabytes of registered memory
    took ~40 s per scan point (measured), and the memory would stay with the rest of the run."""
    cmd = [sys.executable, os.path.abspath(__file__), "--section", name]
    log("section", name, "in a fresh process")
    try:
        pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
        lines = [l for l in pr.stdout.decode().splitlines() if l.strip()]
        if pr.returncode != 0 or not lines:
            return {"error": "section {} exited with {}: {}".format(name, pr.returncode, pr.stderr.decode()[-300:])}
        return json.loads(lines[-1])[name]
    except (subprocess.SubprocessError, OSError, ValueError, KeyError) as exc:
        return {"error": repr(exc)[:300]}


def model_scaling(spec, dev, updates=6):
    """SURVEY 8(e) caveat (ii): until a multi-GPU box exists, 2 / 4 / 8-GPU numbers are MODELLED and labelled so --
    measured single-GPU SGD-step time (the replayed hipGraph of a whole update) at the per-rank shard size of the strict
    mode (320 / N rows) and of the weak mode (320 rows), plus a modelled all-reduce of the flat fp32 gradient per step,
    not overlapped with compute.  Two all-reduce models bracket the answer: a ring over ONE xGMI link pair (what SURVEY
    section 5 prices RCCL's ring at: 2 (N-1)/N S / 153 GB/s + 2 (N-1) hops) and the 2-phase direct exchange over the full
    mesh (reduce-scatter + all-gather, every GPU talking to its N-1 peers at once: 2 (S/N / 153 GB/s + one hop))."""
    from xingtian_amd.model.hip_net import HipActorCritic
    link_gbps, hop_us = 153.0, 2.0
    s_bytes = spec.n_flat * 4
    rng = np.random.default_rng(5)
    # three forms of the SGD step, each the replayed hipGraph of a whole update on THIS GPU:
    #   plain  what N = 1 runs
    #   hook   the data-parallel form behind a generic exchange hook (the RCCL path): row / loss tail in the gradient
    #          buffer, gradient reduction, [hook: nobody to exchange with], squared norm, clip + Adam
    #   fused  the direct exchange FUSED into the step, as a one-rank group: every kernel of the real chain runs against
    #          the rank's own exchange block -- the gradient reduction scatters into the inbox (uncached), one reduce
    #          launch, Adam reads the result buffer.  One rank moves the same 4 x S bytes of uncached traffic through
    #          its device as each of N ranks does (S/N to each of N owners, N inbox slices in, N result slices out, the
    #          whole result read back), so this form already CONTAINS the kernel-side cost of the exchange; only the link
    #          phases are missing.  (Rounds 4-5 added a separately measured three-launch chain on top of the hook form;
    #          with the fused kernels that would count the exchange kernels twice.)
    step_us, step_us_hook, step_us_fused = {}, {}, {}
    from xingtian_amd.parallel import DirectComm
    for rows, form in [(r, f) for r in (320, 160, 80, 40) for f in ("plain", "hook", "fused")]:
        n = ENV_NUM * T_LEN * rows // 320           # the same 52 SGD steps per update at every shard size
        net = HipActorCritic(spec, max_batch=rows, device=str(dev), seed=0)
        one = None
        if form != "plain":
            one = DirectComm(0, 1, int(net.grads_xchg.numel()))
            net.set_dp(0, 1, 1.0)
            if form == "fused":
                one.attach_fused(net)
            else:
                one.attach(net)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        obs = d(rng.integers(0, 256, (n,) + STATE_DIM, dtype=np.uint8))
        act, logp = d(rng.integers(0, A_DIM, n).astype(np.int32)), d((-np.abs(rng.standard_normal(n)) - 0.5).astype(np.float32))
        adv, oldv, tgt = d(rng.standard_normal(n)), d(rng.standard_normal(n).astype(np.float32)), d(rng.standard_normal(n))
        perm = d(np.stack([rng.permutation(n) for _ in range(CFG["NUM_SGD_ITER"])]).astype(np.int32))
        cfg = net.make_ppo_cfg(dict(CFG, BATCH_SIZE=rows), grad_scale=1.0, global_batch=CFG["BATCH_SIZE"])
        for _ in range(3):
            net.ppo_train(cfg, obs, perm, act, logp, adv, oldv, tgt, use_graph=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(updates):
            net.ppo_train(cfg, obs, perm, act, logp, adv, oldv, tgt, use_graph=True)
        torch.cuda.synchronize()
        nsteps = CFG["NUM_SGD_ITER"] * ((n + rows - 1) // rows)
        {"plain": step_us, "hook": step_us_hook, "fused": step_us_fused}[form][rows] = \
            1e6 * (time.perf_counter() - t0) / (updates * nsteps)
        if one is not None:
            assert one.status()["error_bits"] == 0
            if form == "fused":
                one.detach(net)
            else:
                net.lib.xt_net_set_grad_exchange_ex(net.handle, None, None, 0)
            net.set_dp(0, 0)
            one.destroy()
        del net
        torch.cuda.empty_cache()

    def links_us(nr, kind):
        if nr == 1:
            return 0.0
        if kind == "ring_one_link":
            return 2.0 * (nr - 1) / nr * s_bytes / (link_gbps * 1e3) + 2 * (nr - 1) * hop_us
        # direct_2phase: scatter and gather each move (N-1)/N of a slice set over N-1 links at once + one hop each
        return 2.0 * (s_bytes / nr / (link_gbps * 1e3) + hop_us)

    frames_per_update = FRAME_SKIP * ENV_NUM * T_LEN
    sgd_steps = CFG["NUM_SGD_ITER"] * ((ENV_NUM * T_LEN + CFG["BATCH_SIZE"] - 1) // CFG["BATCH_SIZE"])
    base = frames_per_update / (sgd_steps * step_us[320] * 1e-6)
    out = {"MODELLED": "no multi-GPU box was available to the builder: measured 1-GPU time of the data-parallel FORM of the step at the "
                       "shard size + modelled, non-overlapped link phases; NOT a measurement of N GPUs",
           "assumptions": {"allreduce_bytes": s_bytes, "xgmi_link_GBps": link_gbps, "hop_latency_us": hop_us,
                           "ring_one_link": "hook-form step (no exchange kernels) + 2 (N-1)/N S / link + 2 (N-1) hops; RCCL's own "
                                            "kernel time is not priced",
                           "direct_2phase": "fused-form step (the exchange kernels run as a one-rank group against uncached "
                                            "memory: the kernel-side cost is IN the measured step) + 2 (S/N / link + one hop)",
                           "overlap": "none (the two-bucket overlap variants hide the conv backward's ~55 us at 320 rows; not credited)"},
           "measured_sgd_step_us_by_rows": {str(k): round(v, 2) for k, v in step_us.items()},
           "measured_hook_form_step_us_by_rows": {str(k): round(v, 2) for k, v in step_us_hook.items()},
           "measured_fused_form_step_us_by_rows": {str(k): round(v, 2) for k, v in step_us_fused.items()},
           "strict": {}, "weak": {}}
    for nr in (1, 2, 4, 8):
        rows = CFG["BATCH_SIZE"] // nr
        for kind, form in (("ring_one_link", step_us_hook), ("direct_2phase", step_us_fused)):
            ar = links_us(nr, kind)
            # N > 1 runs the data-parallel FORM of the step (measured above), N = 1 the plain one
            t_strict = (form[rows] if nr > 1 else step_us[rows]) + ar
            t_weak = (form[320] if nr > 1 else step_us[320]) + ar
            v_strict = frames_per_update / (sgd_steps * t_strict * 1e-6)
            v_weak = nr * frames_per_update / (sgd_steps * t_weak * 1e-6)
            out["strict"].setdefault(str(nr), {})[kind] = {"value": v_strict, "step_us": round(t_strict, 1), "links_us": round(ar, 1),
                                                           "speedup_vs_1gpu": round(v_strict / base, 2)}
            out["weak"].setdefault(str(nr), {})[kind] = {"value": v_weak, "step_us": round(t_weak, 1), "links_us": round(ar, 1),
                                                         "speedup_vs_1gpu": round(v_weak / base, 2)}
    return out


# ------------------------------------------------------------------------------------------------ main
def main():
    _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--section", default=None, choices=["actor_scan", "env_num_256"],
                    help="DIAGNOSTIC: run only this block of the single-GPU line and print it")
    ap.add_argument("--detail-file", default=None, help="where the full result goes (default: bench_detail.json next to bench.py)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the IMPALA workloads and the plugin-path (e2e) runs")
    ap.add_argument("--quick", action="store_true", help="headline + roofline only (profiling runs)")
    ap.add_argument("--force-dp-path", action="store_true",
                    help="run the N>1 code path (step-wise fwd/bwd -> all-reduce -> clip+Adam) even with one rank")
    ap.add_argument("--dp-variants", default="all",
                    help="data-parallel paths measured next to the step-wise (eager) one when N > 1 or --force-dp-path: 'all' or "
                         "a comma list of hook,hook_overlap,ingraph,ingraph_overlap,direct ('none' = eager only).  Each is validated "
                         "(first update vs eager, replica checksum) before it may carry `value`; failures fall back and are "
                         "reported in dp_variants")
    ap.add_argument("--variant-deadline", type=float, default=120.0,
                    help="seconds a data-parallel variant may take before the watchdog reports the validated ones and exits")
    ap.add_argument("--no-in-graph-stats", action="store_true",
                    help="skip the rocprofv3 --kernel-trace re-run that provides the in-graph per-kernel averages")
    ap.add_argument("--test-backend", default=None, choices=["gloo"],
                    help="DIAGNOSTIC (numbers are meaningless): run the N ranks on however many GPUs are visible (ranks share "
                         "devices, round robin) and exchange gradients through gloo -- exercises the self-spawn and the whole "
                         "N>1 code path on a 1-GPU box, where RCCL refuses two ranks on one device")
    ap.add_argument("--model-scaling", action="store_true",
                    help="print ONLY the modelled 2/4/8-GPU block (measured 1-GPU step time at 40/80/160/320 rows + a modelled "
                         "all-reduce), labelled as modelled")
    ap.add_argument("--workload", default="ppo", choices=["ppo"] + sorted(IMPALA),
                    help="ppo = BASELINE configs[1] (the headline metric, default; its JSON line carries the IMPALA "
                         "workloads as `secondary`); the IMPALA names print that workload's own line (profiling)")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return self_spawn(args)
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise RuntimeError("bench.py --gpus {} is running under a launcher with WORLD_SIZE={}".format(args.gpus, world))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the learner path has no CPU fallback")
    if args.test_backend:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)

    if args.workload != "ppo":
        if world != 1:
            raise RuntimeError("--workload {} is a single-GPU measurement".format(args.workload))
        out = bench_impala(args.workload, args.steps, args.warmup, not (args.no_cpu_baseline or args.quick),
                           in_graph=not (args.quick or args.no_in_graph_stats), quick=args.quick)
        out.update({"n_gpus": 1, "warmup": args.warmup, "higher_is_better": True, "data": "synthetic"})
        return _emit(out)

    if args.section:
        from xingtian_amd.model import netspec as _ns
        fn = {"actor_scan": bench_actor_scan,
              "env_num_256": lambda: bench_env_num_256(_ns.ppo_cnn(STATE_DIM, A_DIM, HIDDEN, "relu", True), torch.device("cuda", local_rank))}
        return _emit({args.section: fn[args.section]()})

    if args.model_scaling:
        from xingtian_amd.model import netspec as _ns
        return _emit({"modelled_scaling": model_scaling(_ns.ppo_cnn(STATE_DIM, A_DIM, HIDDEN, "relu", True),
                                                        torch.device("cuda", local_rank))})

    dist = None
    if world > 1 or args.force_dp_path:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.test_backend:
            dist.init_process_group(args.test_backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from xingtian_amd import lib as L
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    from xingtian_amd.parallel import RcclComm, dp_ppo_update

    dev = torch.device("cuda", local_rank)
    spec = netspec.ppo_cnn(STATE_DIM, A_DIM, HIDDEN, "relu", True)
    lib = L.load()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dp_path = world > 1 or args.force_dp_path
    use_graph = (not dp_path) and not args.no_graph
    bsz = CFG["BATCH_SIZE"]
    perm_rng = np.random.default_rng(1234)   # identical on every rank

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    DP_VARIANTS = ("eager", "hook", "hook_overlap", "ingraph", "ingraph_overlap", "direct")
    comm = {"rccl": None, "direct": None}

    def get_direct():
        """the direct 2-phase all-reduce over peer-mapped memory (xt_allreduce_direct): IPC handles gathered over the
        torch.distributed group; works with the ranks on N GPUs or sharing one"""
        if comm["direct"] is None:
            from xingtian_amd.parallel import DirectComm
            comm["direct"] = DirectComm(rank, world, (spec.n_flat + 3) // 4 * 4 + L.DP_TAIL_FLOATS, timeout_ms=3000).connect()
        return comm["direct"]

    def get_rccl():
        """one raw RCCL communicator for all hook variants (created on first use; its lazy set-up runs outside any
        stream capture)"""
        if comm.get("error"):
            raise RuntimeError(comm["error"])          # the communicator could not be formed: do not retry per variant
        if comm["rccl"] is None:
            try:
                r = RcclComm(rank, world)
            except Exception as exc:                   # noqa: BLE001
                comm["error"] = "raw RCCL communicator unavailable: {!r}".format(exc)
                raise
            warm = torch.zeros(1024, dtype=torch.float32, device=dev)
            r.all_reduce_(warm, L.stream_ptr())
            torch.cuda.synchronize()
            comm["rccl"] = r
        return comm["rccl"]

    def run_mode(mode, variant="eager", steps=None, warmup=None, fixed_perm_seed=None):
        """mode 'weak': rollout seed = rank (own trajectories), full local minibatches; 'strict': the SAME rollout on
        every rank, shards of the global minibatch.  variant (data-parallel path of the weak mode):
          eager            step-wise from Python: fwd/bwd -> torch.distributed all-reduce of the flat gradient -> clip+Adam
          hook             one C call per update (eager enqueue); the library calls ncclAllReduce (raw RCCL communicator)
                           on its own stream between backward and clip+Adam
          hook_overlap     the same with two buckets: last trunk layer + heads all-reduced from a side stream right after
                           the first backward launch, under the conv backward (XT_XCHG_OVERLAP)
          ingraph[_overlap] the same two, captured into the hipGraph of the whole update (no host work per step)
          direct           the 2-phase push exchange over peer-mapped memory (csrc/xt_xgmi.hip) FUSED into the step
                           (xt_net_set_direct), inside the update's hipGraph
        fixed_perm_seed: draw the permutations from a private generator (validation runs: same shuffles for all variants)."""
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        prng = perm_rng if fixed_perm_seed is None else np.random.default_rng(fixed_perm_seed)
        obs, action, logp, value, reward, done = synth_rollout(seed=rank if mode == "weak" else 0)
        n = obs.shape[0]
        net = HipActorCritic(spec, max_batch=bsz, device=str(dev), seed=0)   # same seed -> same replica
        d_obs, d_act, d_logp = d(obs), d(action), d(logp)
        d_value, d_reward, d_done = d(value), d(reward), d(done.astype(np.uint8))
        d_adv = torch.empty((n,), dtype=torch.float64, device=dev)
        d_tgt = torch.empty((n,), dtype=torch.float64, device=dev)
        d_oldv = torch.empty((n,), dtype=torch.float32, device=dev)
        d_perm = torch.empty((CFG["NUM_SGD_ITER"], n), dtype=torch.int32, device=dev)
        if mode == "strict" and variant != "eager":
            # strict sharding INSIDE xt_net_ppo_train (xt_ppo_cfg.shard_rank / shard_world): every rank walks the shared
            # permutations and takes its rows of every global minibatch; means over the global rows, grad_scale 1
            cfg = net.make_ppo_cfg(CFG, grad_scale=1.0, global_batch=0, shard_rank=rank, shard_world=world)
        else:
            cfg = net.make_ppo_cfg(CFG, grad_scale=1.0 / world, global_batch=0)
        rccl = None
        if variant == "direct":
            # (round 6) the exchange FUSED into the step (xt_net_set_dp + xt_net_set_direct): the gradient reduction scatters
            # straight into the owners' inboxes, one reduce launch, Adam reads the exchange block
            rccl = get_direct()
            net.set_dp(rank, world, 1.0 / world if mode == "weak" else 1.0)
            rccl.attach_fused(net)
        elif variant != "eager":
            rccl = get_rccl()
            rccl.attach(net, overlap=variant.endswith("_overlap"))
        graph = (use_graph or variant.startswith("ingraph") or variant == "direct") and not args.no_graph

        # the epoch shuffles travel through a small ring of PINNED blocks (as the product's Model.train does,
        # xingtian_amd/model/ppo/ppo.py::_take_perms): an asynchronous 64 KB DMA instead of a pageable copy that bounces
        # through a staging buffer and a blit kernel; a block is reused only after its copy's event has fired
        perm_pin = [torch.empty((CFG["NUM_SGD_ITER"], n), dtype=torch.int32, pin_memory=True) for _ in range(4)]
        perm_ev = [None] * len(perm_pin)
        perm_turn = [0]

        def new_perms():
            k = perm_turn[0] % len(perm_pin)
            perm_turn[0] += 1
            if perm_ev[k] is not None:
                perm_ev[k].synchronize()
            p = perm_pin[k].numpy()
            inds = np.arange(n)
            for ep in range(CFG["NUM_SGD_ITER"]):
                prng.shuffle(inds)
                p[ep] = inds
            d_perm.copy_(perm_pin[k], non_blocking=True)
            if perm_ev[k] is None:
                perm_ev[k] = torch.cuda.Event()
            perm_ev[k].record(torch.cuda.current_stream(dev))

        def one_update():
            new_perms()
            L.check(lib.xt_gae_f64(L.ptr(d_value), L.ptr(d_reward), L.ptr(d_done), L.ptr(d_adv), L.ptr(d_tgt),
                                   L.ptr(d_oldv), ENV_NUM, T_LEN, 0.99, 0.95, L.stream_ptr()), "gae")
            if not dp_path or rccl is not None:
                net.ppo_train(cfg, d_obs, d_perm, d_act, d_logp, d_adv, d_oldv, d_tgt, use_graph=graph)
            else:
                dp_ppo_update(net, CFG, d_obs, d_perm, d_act, d_logp, d_adv, d_oldv, d_tgt, rank, world, mode=mode)

        for _ in range(warmup):
            one_update()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_update()
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        if variant == "direct":
            st = rccl.status()
            assert st["error_bits"] == 0, "direct all-reduce: a bounded wait ran out: {}".format(st)
            rccl.detach(net)
            net.set_dp(0, 0)
        elif rccl is not None:
            rccl.status(net)
            assert not rccl.errors, "ncclAllReduce failed inside the gradient-exchange hook: {}".format(rccl.errors)
            rccl.detach(net)
        assert torch.isfinite(net.params).all(), "non-finite parameters after the benchmark"
        if dist is not None and world > 1:       # replicas must still be bit-identical
            chk = net.params.double().sum().reshape(1)
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert float(lo) == float(hi), "data-parallel replicas diverged"
        keep = dict(net=net, obs=obs, action=action, logp=logp, value=value, reward=reward, done=done, d_obs=d_obs,
                    d_perm=d_perm, one_update=one_update, rccl=rccl, graph=graph)
        return elapsed, n, keep

    elapsed, n, keep = run_mode("weak")
    dp_variants = {}
    best_variant = "eager"
    if dp_path:
        # ---- the other data-parallel paths, each validated against the step-wise one before it may carry the headline
        dp_variants["eager"] = {"valid": True, "ms_per_step": 1e3 * elapsed / args.steps,
                                "value": FRAME_SKIP * n * world * args.steps / elapsed}
        _, _, kref = run_mode("weak", "eager", steps=1, warmup=0, fixed_perm_seed=99)
        ref = kref["net"].params.clone()
        del kref
        partial = {"done": False}

        def watchdog():
            """a variant that deadlocks (a collective the ranks do not agree on) must not take the measured numbers with
            it: after the deadline rank 0 prints what it has and every rank exits"""
            if partial["done"]:
                return
            log("WATCHDOG: a data-parallel variant did not finish in time; reporting the validated ones")
            if rank == 0:
                cur = dict(partial.get("out") or {})
                cur["dp_variants"] = dict(dp_variants, _watchdog="a variant hung and was abandoned: {}".format(partial.get("running")))
                emit_result(cur)
            os._exit(0)

        import threading
        base = {"metric": "learner env-frames/sec (Atari 84x84x4)", "value": FRAME_SKIP * n * world * args.steps / elapsed,
                "unit": "env-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "fp32", "data": "synthetic", "config": {"workload": "examples/breakout_ppo.yaml PpoCnn (see the full line)",
                                                                 "parallelism": "dp{}".format(world), "dp_mode": "eager"}}
        partial["out"] = base
        for variant in DP_VARIANTS[1:]:
            if args.dp_variants != "all" and variant not in args.dp_variants.split(","):
                continue
            partial["running"] = variant
            timer = threading.Timer(args.variant_deadline, watchdog)
            timer.daemon = True
            timer.start()
            try:
                _, _, kv = run_mode("weak", variant, steps=1, warmup=0, fixed_perm_seed=99)
                diff = float((kv["net"].params - ref).abs().max())
                scale = float(ref.abs().max())
                bitwise = bool(torch.equal(kv["net"].params, ref))
                del kv
                if not (diff <= 1e-5 * max(scale, 1e-30)):
                    raise AssertionError("first update differs from the step-wise path: max |d| {:.3e} (scale {:.3e})".format(diff, scale))
                el, n_v, kv = run_mode("weak", variant)
                dp_variants[variant] = {"valid": True, "ms_per_step": 1e3 * el / args.steps,
                                        "value": FRAME_SKIP * n_v * world * args.steps / el,
                                        "first_update_max_abs_diff_vs_eager": diff, "first_update_bitwise_equal": bitwise,
                                        "hip_graph": bool(kv["graph"]),
                                        "ranks": world if variant == "direct" else get_rccl().count()}
                if el < elapsed:
                    elapsed, keep, best_variant = el, kv, variant
                else:
                    del kv
            except Exception as exc:      # noqa: BLE001 -- fall back to the paths that did validate, and say so
                dp_variants[variant] = {"valid": False, "error": repr(exc)[:300]}
                try:
                    if comm["rccl"] is not None:
                        comm["rccl"].errors.clear()
                except Exception:         # noqa: BLE001
                    pass
            finally:
                timer.cancel()
            torch.cuda.empty_cache()
        partial["done"] = True
    frames = FRAME_SKIP * n * world * args.steps
    sgd_steps = CFG["NUM_SGD_ITER"] * ((n + bsz - 1) // bsz)
    graph_on = bool(keep["graph"])
    out = {
        "metric": "learner env-frames/sec (Atari 84x84x4)", "value": frames / elapsed, "unit": "env-frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "examples/breakout_ppo.yaml PpoCnn 84x84x4 uint8, env_num=32/GPU, T=128, "
                               "BATCH_SIZE=320/GPU, NUM_SGD_ITER=4, hidden 256, A=4; step = GAE + full PPO update "
                               "(52 SGD steps) on an HBM-resident rollout",
                   "env_steps_per_update": n * world, "sgd_steps_per_update": sgd_steps,
                   "global_batch": bsz * world, "parallelism": "dp{}".format(world), "hip_graph": graph_on,
                   "dp_path": bool(dp_path)},
    }
    if args.detail_file:
        out["detail_file"] = os.path.abspath(args.detail_file)
    if args.test_backend:
        out["config"]["DIAGNOSTIC"] = "ranks share GPUs, gradients through {}: not a measurement".format(args.test_backend)
    if dp_path:
        out["config"]["dp_mode"] = best_variant
        out["dp_variants"] = dict(dp_variants, note="`value` is the fastest variant whose first update matched the step-wise "
                                  "(eager) path within 1e-5 of the parameter scale and whose replicas stayed bit-identical; "
                                  "variants that failed or were skipped say so")
        out["config"]["collective"] = "all-reduce (SUM) of the flat fp32 gradient ({} floats) per SGD step; eager: torch.distributed {}; "                                       "hook / ingraph: raw RCCL inside xt_net_ppo_train (one bucket, or two with XT_XCHG_OVERLAP)".format(
            spec.n_flat, args.test_backend or "nccl (RCCL)")
        if dist is not None:
            out["config"]["ranks_in_group"] = dist.get_world_size()
    out["update_tflops"] = PPO_MFLOP_PER_SAMPLE_PASS * 1e6 * n * CFG["NUM_SGD_ITER"] * world * args.steps / elapsed / 1e12

    if world > 1:
        # ---- strict data parallelism: the reference's global minibatch of 320 rows sharded over the ranks
        keep_weak = keep
        del keep
        keep_weak.pop("one_update"); keep_weak.pop("net")
        torch.cuda.empty_cache()
        try:
            el_s, n_s, keep_s = run_mode("strict")
            strict_variants = {"eager": {"valid": True, "ms_per_step": 1e3 * el_s / args.steps}}
            ref_s = None
            for variant in ("hook", "ingraph", "direct"):
                if args.dp_variants != "all" and variant not in args.dp_variants.split(","):
                    continue
                try:
                    if ref_s is None:
                        _, _, kr = run_mode("strict", "eager", steps=1, warmup=0, fixed_perm_seed=99)
                        ref_s = kr["net"].params.clone()
                        del kr
                    _, _, kv = run_mode("strict", variant, steps=1, warmup=0, fixed_perm_seed=99)
                    diff = float((kv["net"].params - ref_s).abs().max())
                    bitwise = bool(torch.equal(kv["net"].params, ref_s))
                    del kv
                    if not (diff <= 1e-5 * max(float(ref_s.abs().max()), 1e-30)):
                        raise AssertionError("first strict update differs from the step-wise path: max |d| {:.3e}".format(diff))
                    el_v, _, kv = run_mode("strict", variant)
                    del kv
                    strict_variants[variant] = {"valid": True, "ms_per_step": 1e3 * el_v / args.steps,
                                                "first_update_bitwise_equal": bitwise}
                    if el_v < el_s:
                        el_s = el_v
                except Exception as exc:      # noqa: BLE001
                    strict_variants[variant] = {"valid": False, "error": repr(exc)[:300]}
                torch.cuda.empty_cache()
            out["strict"] = {"value": FRAME_SKIP * n_s * args.steps / el_s, "unit": "env-frames/s", "scaling": "strong",
                             "ms_per_step": 1e3 * el_s / args.steps, "global_batch": bsz, "dp_variants": strict_variants,
                             "rows_per_gpu": bsz // world, "env_steps_per_update": n_s, "sgd_steps_per_update": sgd_steps,
                             "note": "same rollout + same permutations on every rank, rank r takes rows "
                                     "[r*B/N, (r+1)*B/N) of every global minibatch; loss means over the global minibatch; "
                                     "arithmetic of the single-GPU update up to fp32 summation order"}
            del keep_s
        except Exception as exc:      # noqa: BLE001 -- the weak-mode headline above must still be reported
            out["strict"] = {"error": repr(exc)}
        torch.cuda.empty_cache()
        try:
            out["secondary"] = [bench_impala_dp(k, rank, world, dev, dist) for k in ("pong_impala_speedup", "breakout_impala")]
        except Exception as exc:      # noqa: BLE001
            out["secondary"] = [{"error": repr(exc)}]
        if rank == 0:
            #  the N > 1 line carries `roofline` and `cpu_baseline` too: the dominant kernel timed on THIS
            # rank's GPU at the per-rank minibatch of the headline (weak mode: BATCH_SIZE rows; isolated launches -- a
            # rocprofv3 re-run of an N-rank job is not attempted) and the CPU restatement on this box's host cores
            try:
                netr = HipActorCritic(spec, max_batch=bsz, device=str(dev), seed=0)
                kern = layer_rooflines(netr, spec, bsz, keep_weak["d_obs"], keep_weak["d_perm"][0, :bsz].contiguous(), x6=True)
                out["roofline"] = roofline_of(kern, "ppo", None)
                out["roofline"]["rows_per_gpu"] = bsz
                out["library"] = library_identity()
                del netr
            except Exception as exc:      # noqa: BLE001
                out["roofline_error"] = repr(exc)[:200]
            if not (args.no_cpu_baseline or args.quick):
                try:
                    out["cpu_baseline"] = cpu_baseline_ppo(*(keep_weak[k] for k in ("obs", "action", "logp", "value", "reward", "done")),
                                                           max_seconds=9.0)
                except Exception as exc:      # noqa: BLE001
                    out["cpu_baseline_error"] = repr(exc)[:200]
            emit_result(out)
        dist.barrier()
        dist.destroy_process_group()
        return

    # ------------------------------------------------------------------ single GPU: the rest of the line (rank 0)
    net, d_obs, d_perm = keep["net"], keep["d_obs"], keep["d_perm"]
    idx = d_perm[0, :bsz].contiguous()
    kern = layer_rooflines(net, spec, bsz, d_obs, idx, x6=True)
    ig, ig_note = (None, "skipped") if (args.quick or args.no_in_graph_stats or dp_path) else in_graph_kernel_stats("ppo")
    out["roofline"] = roofline_of(kern, "ppo", ig)
    out["roofline"]["in_graph_source"] = ig_note
    out["library"] = library_identity()
    out["box"] = box_health(out["roofline"])
    out["degraded_box"] = out["box"]["degraded_box"]
    if not args.quick:
        # sustained: the same loop for >= 2 s (the K-step region above is ~0.16 s at K=20)
        one_update = keep["one_update"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 2.0:
            for _ in range(10):
                one_update()
            torch.cuda.synchronize()
            reps += 10
        el = time.perf_counter() - t0
        out["sustained"] = {"seconds": el, "updates": reps, "value": FRAME_SKIP * n * reps / el,
                            "ms_per_step": 1e3 * el / reps}

        def busy():
            for _ in range(10):
                one_update()
            torch.cuda.synchronize()
        out["device"] = device_report(busy)
        out["box"].update({k: out["device"].get(k) for k in ("sclk_mhz_busy", "power_w_busy", "power_cap_w") if k in out["device"]})
    obs, action, logp, value, reward, done = (keep[k] for k in ("obs", "action", "logp", "value", "reward", "done"))
    del keep, net
    torch.cuda.empty_cache()
    if not (args.quick or args.no_secondary):
        out["e2e"] = {"definition": "SURVEY 8(d): env-steps of one Algorithm.train() / wall time of prepare_data x k + "
                                    "train() (incl. H2D of the uint8 rollout) + get_weights() (D2H), plugin classes",
                      "env_num_32": bench_e2e_ppo(32), "env_num_10_yaml": bench_e2e_ppo(10),
                      "env_num_32_learner_gae": bench_e2e_ppo(32, learner_gae=True),
                      "env_num_32_publish": bench_e2e_ppo(32, handover="publish"),
                      "env_num_32_pinned_ring": bench_e2e_ppo(32, via_ring=True, handover="publish")}
        from xingtian_amd import ingest
        out["e2e"]["staging_copy"] = dict(ingest.staging_report(), note="xt_stage_tune on this host: GB/s of the pageable -> "
                                          "pinned copy per variant (memcpy / non-temporal stores x inline,1,2,4,8 worker "
                                          "threads, 4 MiB pieces); the fastest is what prepare_data uses")
        out["value_e2e"] = out["e2e"]["env_num_32"]["value"]
        for key in ("env_num_256", "actor_scan"):
            out[key] = run_section(key)
        try:
            out["modelled_scaling"] = model_scaling(spec, dev)
        except Exception as exc:      # noqa: BLE001 -- a diagnostic block must not take the line with it
            out["modelled_scaling"] = {"error": repr(exc)[:300]}
        out["secondary"] = [bench_impala(k, 10, 3, not args.no_cpu_baseline and "_" not in k[len("breakout_impala"):],
                                         in_graph=not args.no_in_graph_stats and k in ("breakout_impala", "pong_impala_speedup"))
                            for k in ("breakout_impala", "pong_impala_speedup", "breakout_impala_batched",
                                      "pong_impala_per_message")]
    if not (args.no_cpu_baseline or args.quick):
        out["cpu_baseline"] = cpu_baseline_ppo(obs, action, logp, value, reward, done)
    emit_result(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
