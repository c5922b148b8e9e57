#!/usr/bin/env python
"""First minutes on a multi-GPU node: a verdict on the data-parallel learner's cross-device assumptions instead of a hang.

Everything data parallel in this repository has only ever run with all ranks on ONE MI355X (gpurun boxes have one GPU): RCCL
with a one-rank communicator, the direct 2-phase exchange (csrc/xt_xgmi.hip) with up to eight processes sharing the device.
What has never met hardware: uncached peer writes over xGMI + `s_waitcnt vmcnt(0)` before the system-scope flag store as the
ONLY ordering of the direct exchange, RCCL collectives inside a replayed hipGraph, RCCL with more than one rank.  On any box
with >= 2 visible devices this script (< 60 s, every wait bounded) runs, one process per GPU:

  1. the hipDeviceCanAccessPeer matrix;
  2. a raw RCCL communicator: ncclCommCount == N, an all-reduce checked against torch.distributed;
  3. xt_direct_create / connect across devices (hipIpc), 200 value-checked all-reduces (fused single launch and the
     three-launch chain, both flat-gradient sizes, an odd tail) -- bitwise the host's rank-order float32 sum AND within
     rounding of torch.distributed's; xt_direct_info must report ONE rank per device;
  4. one strict and one weak PPO update through the fused direct exchange inside a replayed hipGraph and through RCCL in the
     graph, validated against the step-wise path (fwd/bwd -> torch.distributed all-reduce -> clip + Adam), replicas bitwise
     equal across the ranks.

    python tools/multi_gpu_preflight.py            # all visible GPUs
    python tools/multi_gpu_preflight.py --gpus 4

Prints one JSON line {"ok": bool, "world": N, "checks": {...}}; exit code 0 iff ok.  `tests/test_gpu_preflight.py` runs it under
`-m gpu` and skips on a one-GPU box.  Reference analogue: the dead host-side exchange, xt/framework/trainer.py:86-92."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher(args):
    import torch
    n_dev = torch.cuda.device_count()
    world = args.gpus or n_dev
    if args.one_gpu_selftest:
        # the SCRIPT's own logic exercised on a one-GPU box: `--gpus N` ranks share device 0, torch.distributed over gloo, the
        # RCCL checks are skipped (RCCL refuses several ranks on one device) -- so that the first run on a real multi-GPU node
        # does not die of a typo in this file
        world = max(2, args.gpus or 2)
        peer = [[True] * world for _ in range(world)]
    elif n_dev < 2 or world < 2:
        print(json.dumps({"ok": False, "skipped": "needs >= 2 visible GPUs, found {}".format(n_dev)}))
        return 2
    elif world > n_dev:
        print(json.dumps({"ok": False, "error": "--gpus {} > {} visible devices".format(world, n_dev)}))
        return 1
    else:
        peer = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(world)] for i in range(world)]
    out = os.path.join(args.outdir, "preflight_{}.json".format(os.getpid()))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), "--worker", "--out", out]
    if args.one_gpu_selftest:
        cmd.append("--one-gpu-selftest")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    t0 = time.time()
    try:
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=args.timeout)
        rc, log = proc.returncode, proc.stdout.decode()[-4000:]
    except subprocess.TimeoutExpired as exc:
        rc, log = -1, "TIMEOUT after {} s\n".format(args.timeout) + (exc.stdout or b"").decode()[-4000:]
    res = {"ok": False, "world": world, "peer_access": peer, "wall_s": round(time.time() - t0, 1)}
    if os.path.exists(out):
        with open(out) as f:
            res.update(json.load(f))
        os.remove(out)
    res["ok"] = bool(rc == 0 and res.get("checks") and all(v.get("ok") for v in res["checks"].values())
                     and all(all(r) for r in peer))
    if not res["ok"]:
        res["log_tail"] = log
    print(json.dumps(res))
    return 0 if res["ok"] else 1


def worker(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    one_gpu = bool(args.one_gpu_selftest)
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("gloo" if one_gpu else "nccl", rank=rank, world_size=world)
    from xingtian_amd import lib as L
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    from xingtian_amd.parallel import DirectComm, RcclComm, dp_ppo_update
    checks = {}

    def record(name, fn):
        t0 = time.time()
        try:
            extra = fn() or {}
            checks[name] = dict(ok=True, s=round(time.time() - t0, 2), **extra)
        except Exception as exc:      # noqa: BLE001
            checks[name] = dict(ok=False, s=round(time.time() - t0, 2), error=repr(exc)[:500])
        flag = torch.tensor([1 if checks[name]["ok"] else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # (a check fails everywhere if it fails anywhere)
        if not int(flag.item()) and checks[name]["ok"]:
            checks[name] = dict(ok=False, error="failed on another rank")
        return checks[name]["ok"]

    # ---- 2. raw RCCL
    state = {}

    def rccl_check():
        r = state["rccl"] = RcclComm(rank, world)
        assert r.count() == world, "ncclCommCount {} != {}".format(r.count(), world)
        x = torch.arange(4096, dtype=torch.float32, device=dev) * (rank + 1)
        y = x.clone()
        r.all_reduce_(x, L.stream_ptr())
        dist.all_reduce(y)
        torch.cuda.synchronize()
        assert torch.equal(x, y), "raw ncclAllReduce disagrees with torch.distributed"
        return {"ranks": r.count()}

    if not one_gpu:
        record("rccl_raw_communicator", rccl_check)

    # ---- 3. the direct exchange across devices
    counts = [847496 + 32, 1005112 + 32, 4099, 7]

    def direct_check():
        c = state["direct"] = DirectComm(rank, world, max(counts), timeout_ms=5000).connect()
        info = c.info()
        assert info["ranks_on_device"] == (world if one_gpu else 1), "xt_direct_info: {} ranks on this device ({} expected)".format(
            info["ranks_on_device"], world if one_gpu else 1)
        n_done = 0
        for fused in (True, False):
            c.set_fused(fused)
            for it in range(100):
                cnt = counts[it % len(counts)]
                gen = lambda r: np.random.default_rng(1000 * it + r).standard_normal(cnt).astype(np.float32)   # noqa: E731
                buf = torch.from_numpy(gen(rank)).to(dev)
                ref = buf.clone()
                c.all_reduce_(buf)
                dist.all_reduce(ref)
                torch.cuda.synchronize()
                want = gen(0)
                for r in range(1, world):
                    want = (want + gen(r)).astype(np.float32)
                got = buf.cpu().numpy()
                assert np.array_equal(got, want), "direct all-reduce {} (count {}, fused {}): max |diff| vs the rank-order host sum {}".format(
                    it, cnt, fused, float(np.abs(got - want).max()))
                assert np.allclose(got, ref.cpu().numpy(), rtol=1e-5, atol=1e-5), "direct all-reduce disagrees with torch.distributed"
                n_done += 1
        st = c.status()
        assert st["error_bits"] == 0, "error bits {}".format(st["error_bits"])
        return {"all_reduces": n_done, "block_cap": info["block_cap"]}

    record("direct_exchange_200_value_checked", direct_check)

    # ---- 4. one strict + one weak update, fused direct / RCCL in the graph against the step-wise path
    cfg = dict(LR=2.5e-4, LOSS_CLIPPING=0.1, ENTROPY_LOSS=0.003, VF_CLIP=5.0, CRITIC_LOSS_COEF=1.0, MAX_GRAD_NORM=5.0,
               BATCH_SIZE=64, NUM_SGD_ITER=2)
    spec = netspec.ppo_cnn((84, 84, 4), 4, (256,), "relu", True)
    n = 160
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731

    def rollout(seed):
        rng = np.random.default_rng(seed)
        obs = rng.integers(0, 256, (n, 84, 84, 4)).astype(np.uint8)
        return (d(obs), d(rng.integers(0, 4, n).astype(np.int32)), d((-np.abs(rng.standard_normal(n)) - 0.5).astype(np.float32)),
                d(rng.standard_normal(n)), d(rng.standard_normal(n).astype(np.float32)), d(rng.standard_normal(n)))

    perm = d(np.stack([np.random.default_rng(5).permutation(n) for _ in range(2)]).astype(np.int32))

    def update_check(mode, exchange):
        def body():
            obs, act, logp, adv, oldv, tgt = rollout(100 if mode == "strict" else 200 + rank)
            ref = HipActorCritic(spec, max_batch=64, device=str(dev), seed=0)
            dp_ppo_update(ref, cfg, obs, perm, act, logp, adv, oldv, tgt, rank, world, mode=mode)
            torch.cuda.synchronize()
            net = HipActorCritic(spec, max_batch=64, device=str(dev), seed=0)
            start = net.params.clone()
            if mode == "strict":
                c = net.make_ppo_cfg(cfg, grad_scale=1.0, global_batch=0, shard_rank=rank, shard_world=world)
            else:
                c = net.make_ppo_cfg(cfg, grad_scale=1.0 / world, global_batch=0)
            net.set_dp(rank, world, 1.0 if mode == "strict" else 1.0 / world)
            if exchange == "direct":
                comm = DirectComm(rank, world, int(net.grads_xchg.numel()), timeout_ms=5000).connect()
                comm.attach_fused(net)
            else:
                comm = state["rccl"]
                comm.attach(net)
            for rep in range(2):            # capture, then REPLAY from the same start
                net.params.copy_(start); net.reset_optimizer()
                acc = net.ppo_train(c, obs, perm, act, logp, adv, oldv, tgt, use_graph=True)
                a = net.read_loss(acc)
            comm.detach(net)
            net.set_dp(0, 0)
            err = float((net.params - ref.params).norm() / (ref.params - start).norm())
            assert err < 5e-3, "{} update through {} differs from the step-wise path: {}".format(mode, exchange, err)
            chk = net.params.double().sum().reshape(1)
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert float(lo) == float(hi), "replicas diverged"
            if exchange == "direct":
                assert comm.status()["error_bits"] == 0
                comm.destroy()
            return {"rel_err_vs_stepwise": err, "loss": float(a[0] / max(a[1], 1.0))}
        return body

    for mode in ("strict", "weak"):
        for exchange in (("direct",) if one_gpu else ("direct", "rccl")):
            record("update_{}_{}_in_graph".format(mode, exchange), update_check(mode, exchange))

    dist.barrier()
    if rank == 0:
        with open(args.out, "w") as f:
            json.dump({"checks": checks}, f)
    ok = all(v["ok"] for v in checks.values())
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0)
    ap.add_argument("--timeout", type=int, default=240)
    ap.add_argument("--outdir", default="/tmp")
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--one-gpu-selftest", action="store_true",
                    help="exercise this script on a ONE-GPU box: the ranks share device 0 over gloo, RCCL checks skipped")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    sys.exit(worker(args) if args.worker else launcher(args))


if __name__ == "__main__":
    main()
