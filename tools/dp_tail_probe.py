#!/usr/bin/env python
"""Kernel-side cost of the data-parallel step's TAIL (what follows the backward pass) on ONE device (C ABI xt_net_time_tail):

  plain    gradient reduction + clip/Adam                                  (the single-GPU step: 2 launches)
  fused    xt_net_set_dp + xt_net_set_direct: gradient reduction scattering into the owners' inboxes -> one reduce launch
           (fixed rank order, squared-norm partials) -> Adam reading the exchange block       (3 launches, no copies)

    python tools/dp_tail_probe.py              # one process: plain vs the fused chain as a one-rank group
    python tools/dp_tail_probe.py --procs 8    # N processes sharing the GPU, all ranks' traffic on that one device: the time
                                               # of one tail for ALL ranks; / N = per-rank kernel-side cost

PpoCnn 84x84x4 (847 496 parameters, 3.39 MB) after one B-row gradient step.  The xGMI phases come on top on a real node."""
import argparse
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    from xingtian_amd.parallel import DirectComm
    rows = args.rows
    spec = netspec.ppo_cnn((84, 84, 4), 4, (256,), "relu", True)
    net = HipActorCritic(spec, max_batch=rows, seed=0)
    rng = np.random.default_rng(3)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
    cfg = dict(LR=2.5e-4, LOSS_CLIPPING=0.1, ENTROPY_LOSS=0.003, VF_CLIP=5.0, CRITIC_LOSS_COEF=1.0, MAX_GRAD_NORM=5.0,
               BATCH_SIZE=rows, NUM_SGD_ITER=1)
    obs = d(rng.integers(0, 256, (rows, 84, 84, 4)).astype(np.uint8))
    net.ppo_step(net.make_ppo_cfg(cfg), obs, None, d(rng.integers(0, 4, rows).astype(np.int32)),
                 d((-np.abs(rng.standard_normal(rows)) - 0.5).astype(np.float32)), d(rng.standard_normal(rows)),
                 d(rng.standard_normal(rows).astype(np.float32)), d(rng.standard_normal(rows)), apply=False)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    out = {}
    for rep in range(args.repeats):
        barrier()
        out.setdefault("plain_us", []).append(1e3 * net.time_tail(reps=args.reps))
    comm = DirectComm(rank, world, int(net.grads_xchg.numel()), timeout_ms=5000)
    if world > 1:
        comm.connect()
    net.set_dp(rank, world, 1.0)
    comm.attach_fused(net)
    for rep in range(args.repeats):
        barrier()
        out.setdefault("fused_us", []).append(1e3 * net.time_tail(reps=args.reps))
    barrier()
    st = comm.status()
    assert st["error_bits"] == 0, st
    out["info"] = comm.info()
    comm.detach(net)
    net.set_dp(0, 0)
    comm.destroy()
    res = {k: (round(float(np.median(v)), 2) if isinstance(v, list) else v) for k, v in out.items()}
    res.update(world=world, rows=rows)
    if world > 1:
        allres = [None] * world
        dist.all_gather_object(allres, res)
        if rank == 0:
            plain = max(r["plain_us"] for r in allres)
            fused = max(r["fused_us"] for r in allres)
            print(json.dumps({"procs": world, "rows": rows, "plain_tail_us_all_ranks": plain, "fused_tail_us_all_ranks": fused,
                              "plain_us_per_rank": round(plain / world, 2), "fused_us_per_rank": round(fused / world, 2),
                              "exchange_us_per_rank": round((fused - plain) / world, 2), "info": res["info"]}))
        dist.barrier()
        dist.destroy_process_group()
    else:
        print(json.dumps(dict(res, exchange_us=round(res["fused_us"] - res["plain_us"], 2))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--rows", type=int, default=40)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--worker", action="store_true")
    args = ap.parse_args()
    if args.worker or args.procs <= 1:
        return worker(args)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.procs), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), "--worker", "--rows", str(args.rows),
           "--reps", str(args.reps), "--repeats", str(args.repeats)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    for line in proc.stdout.decode().splitlines():
        if line.startswith("{"):
            print(line)
    if proc.returncode != 0:
        print(proc.stdout.decode()[-3000:])
    return proc.returncode


if __name__ == "__main__":
    sys.exit(main() or 0)
