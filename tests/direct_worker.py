"""Rank program of tests/test_gpu_direct.py: N PROCESSES on the one visible GPU run the direct 2-phase all-reduce
(xt_allreduce_direct) over hipIpc-mapped exchange blocks -- the code path N GPUs of one node take (SURVEY.md 8(e) caveat i).
The 64-byte IPC handles travel through a gloo group.  Every rank checks its result BITWISE against the fixed-rank-order
float32 sum computed on the host, over batches of back-to-back all-reduces (no host synchronisation inside a batch: a stale
read of an inbox / result buffer or a lost flag would show)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

COUNTS = (847496, 1005109, 4099, 64, 1, 250007, 3, 8)      # PpoCnn / ImpalaCnnOpt flat sizes, odd tails, tiny


def rank_input(it, r, count):
    base = np.random.default_rng(1000 + it).standard_normal(count).astype(np.float32)
    return (base * np.float32(1.0 + 0.37 * r) + np.float32(0.001 * r * (it + 1))).astype(np.float32)


def expected_sum(it, world, count):
    acc = rank_input(it, 0, count)
    for p in range(1, world):
        acc = acc + rank_input(it, p, count)          # float32, rank order 0..N-1: what the reduce kernel does
    return acc


def main():
    outdir, iters, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    mode = sys.argv[4] if len(sys.argv) > 4 else "fused"          # "fused": one launch per all-reduce; "chain": three
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from xingtian_amd.parallel import DirectComm
    comm = DirectComm(rank, world, max(COUNTS), timeout_ms=30000).connect().set_fused(mode == "fused")
    it = 0
    bad = []
    while it < iters:
        todo = list(range(it, min(iters, it + batch)))
        bufs = [torch.from_numpy(rank_input(i, rank, COUNTS[i % len(COUNTS)])).cuda() for i in todo]
        torch.cuda.synchronize()
        for b in bufs:                         # back to back, no host sync in between
            comm.all_reduce_(b)
        torch.cuda.synchronize()
        for i, b in zip(todo, bufs):
            want = expected_sum(i, world, COUNTS[i % len(COUNTS)])
            if not np.array_equal(b.cpu().numpy(), want):
                bad.append(i)
        it += len(todo)
    st = comm.status()
    ok = (not bad) and st["error_bits"] == 0 and st["seq"] == iters and st["calls"] == iters
    # timing (all ranks share ONE GPU: the kernel-side chain + the local traffic of N ranks, not xGMI)
    import time
    buf = torch.from_numpy(rank_input(0, rank, COUNTS[0])).cuda()
    for _ in range(5):
        comm.all_reduce_(buf)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(100):
        comm.all_reduce_(buf)
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / 100
    with open(os.path.join(outdir, "direct_r{}.txt".format(rank)), "w") as f:
        f.write("{} bad={} status={} mode={} us_per_allreduce_{}_floats={:.1f}\n".format("OK" if ok else "FAIL", bad[:10], st, mode,
                                                                                       COUNTS[0], us))
    dist.barrier()
    comm.destroy()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
